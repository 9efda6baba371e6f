/*
 * fxjps.h -- C ABI of libfxjps.so: batched Jump-Point-Search A* on MI355X (gfx950).
 *
 * Drop-in boundary for the one hot path of fuxi-planner, the grid search
 *     jps1.method(matrix, start, goal, hchoice)            scripts/jps1.py:183-230
 * called once per planner tick from
 *     scripts/global_planner_st.py:285 and scripts/global_planner_ccst.py:477.
 * The Python shim (fuxi-planner_amd/jps1.py, ctypes) is the only thing the ROS
 * node sees; everything below is what that shim binds.  Plain pointers and
 * sizes only -- no C++ types, no torch types, no exceptions cross this ABI.
 *
 * Conventions (same as the reference):
 *   - grid is W x H, indexed matrix[x][y], row-major with y contiguous
 *     (occ[x*H + y]); a cell is an obstacle iff its byte is non-zero (the shim
 *     converts with `matrix == 1`, jps1.py:20-29);
 *   - cells are (x, y) int32 pairs; a path is the list of JUMP POINTS from
 *     start to goal inclusive, exactly the list jps1.method returns
 *     (jps1.py:200-205), and its cost is the float64 it prints (jps1.py:207);
 *   - hchoice 1 = octile x10/x14, 2 = Euclidean (jps1.py:3-12, 232-246).
 *
 * Ownership: the caller owns every host buffer it passes; the library never
 * keeps a pointer past the call.  The handle owns all device memory, streams
 * and communicators.  Threading: a handle is not thread-safe; distinct handles
 * are independent.  All calls block until their results are in host memory.
 *
 * Errors: every int-returning function returns FXJPS_OK (0) or a negative
 * code; fxjps_last_error() then describes it.  Per-query conditions never
 * abort a batch: they are reported in out_len[q].
 */
#ifndef FXJPS_H
#define FXJPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions declared between this push and the pop at the end of the
 * header are its whole dynamic symbol table (tests/test_host_cpu.py checks `nm -D` against this list, both ways). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct fxjps fxjps_t;

/* library-level return codes */
#define FXJPS_OK 0
#define FXJPS_E_ARG (-1)     /* bad argument (NULL, sizes, hchoice not in {1,2}: jps1.py:188 TypeError) */
#define FXJPS_E_NODEV (-2)   /* no usable HIP device: the planner has no CPU fallback */
#define FXJPS_E_HIP (-3)     /* a HIP runtime call failed */
#define FXJPS_E_NOGRID (-4)  /* plan before set_grid */
#define FXJPS_E_NOMEM (-5)
#define FXJPS_E_COMM (-6)    /* RCCL failure (multi-device handle) */

/* per-query codes in out_len[q] */
#define FXJPS_Q_NOPATH 0           /* jps1.method returned (0, t): jps1.py:230 */
#define FXJPS_Q_PATH_TOO_LONG (-1) /* more than max_path_len jump points; cost is still valid */
#define FXJPS_Q_BAD_START (-2)     /* start outside the grid: the reference raises IndexError */
#define FXJPS_Q_CAPACITY (-3)      /* search state outgrew device scratch even after the large-scratch retry */

/* backend ids for fxjps_create.  Only the HIP backend exists.  SURVEY.md 8(b) sketched "0 = CPU, 1 = HIP";
 * id 0 is deliberately NOT implemented and fxjps_create(0, ...) returns FXJPS_E_ARG: a CPU backend inside the product
 * would be a silent fallback for the very path this library exists to run on the GPU (the only CPU implementation
 * in the repository is the test oracle under oracle/, which the library never links or loads). */
#define FXJPS_BACKEND_HIP 1

/* Version of this header: fxjps_version() of the loaded library must equal it.  History of what a C host has to know:
 *   300  round 3
 *   300  (round 4, NOT bumped -- a mistake) fxjps_timing_t grew by head_launch_ms, batch_launch_ms, solo_timeouts
 *   500  round 5: the version says so now; fxjps_timing_size() / fxjps_last_timing_sized() for hosts that want to be safe
 *        against the next growth; the whole-grid setters refuse handles of fxjps_create_rank with world > 1;
 *        fxjps_rank_preflight, fxjps_reserve_grid.
 *   600  round 6: fxjps_get_grid_context (the resident grid of ONE context of a multi-device handle: what a host compares
 *        after the broadcast); the library exports nothing but the functions of this header (-fvisibility=hidden + an
 *        export map); the text of a failed fxjps_create* is kept per calling thread.
 * fxjps_timing_t only ever grows at its end. */
#define FXJPS_VERSION 600
int fxjps_version(void);

/* Number of HIP devices visible, or a negative code. */
int fxjps_device_count(void);

/* Create a planner on the given devices (device_ids == NULL: devices 0..n_dev-1).
 * n_dev > 1 shards every batch over the devices in contiguous slices and
 * broadcasts the grid from device_ids[0] with RCCL.  backend must be
 * FXJPS_BACKEND_HIP; there is no CPU backend. */
int fxjps_create(int backend, const int* device_ids, int n_dev, fxjps_t** out);

/* One process per GPU without any framework in the process (north_star: "no PyTorch"; the reference shares nothing
 * between calls, jps1.py:183-192, so the ranks need only the grid).  Rank 0 asks RCCL for a communicator id
 * (ncclGetUniqueId: 128 bytes) and hands it to the other ranks by whatever channel the launcher has -- bench.py and
 * fuxi_planner_amd.ranks use a TCP socket at MASTER_ADDR:MASTER_PORT+1; every rank then creates its handle on its own
 * device with fxjps_create_rank (ncclCommInitRank: collective, all ranks call it), and every rank calls
 * fxjps_set_grid_rank with the same W, H -- rank 0 with the occupancy bytes, the others with NULL: ONE ncclBroadcast of
 * W*H bytes over xGMI, then each rank builds its maps itself and plans its own contiguous slice of the queries
 * (fxjps_plan_batch* as on any handle).  world == 1 needs no id and makes no communicator. */
int fxjps_rank_unique_id(void* out_id128);
int fxjps_create_rank(int device, int rank, int world, const void* id128, fxjps_t** out);
int fxjps_set_grid_rank(fxjps_t* h, const uint8_t* occ, int32_t W, int32_t H);
/* fxjps_create_rank (ncclCommInitRank) and fxjps_set_grid_rank (ncclBroadcast) are COLLECTIVES: a rank that fails before
 * it joins one leaves the others waiting inside RCCL.  What can fail on ONE rank is therefore checked first, without any
 * collective, and the ranks agree on the outcome over their own channel before anybody enters (fuxi_planner_amd.ranks does):
 * fxjps_rank_preflight -- the device exists and takes an allocation, librccl loads and has the four entry points used;
 * fxjps_reserve_grid   -- the handle's buffers for a W x H grid are allocated (what fxjps_set_grid_rank would allocate).
 * On a handle of fxjps_create_rank with world > 1 every OTHER whole-grid setter (fxjps_set_grid, _device, _image,
 * fxjps_prepare_*) returns FXJPS_E_ARG: it would enter a broadcast alone. */
int fxjps_rank_preflight(int device);
int fxjps_reserve_grid(fxjps_t* h, int32_t W, int32_t H);

void fxjps_destroy(fxjps_t* h);

/* Last error text for this handle (h == NULL: last create error). */
const char* fxjps_last_error(fxjps_t* h);

/* Upload an occupancy grid (copies; replaces the previous one) and rebuild the
 * derived device maps.  Replaces the `matrix` argument of jps1.method. */
int fxjps_set_grid(fxjps_t* h, const uint8_t* occ, int32_t W, int32_t H);

/* Same, but `d_occ` already lives in the memory of the handle's first device
 * (e.g. the receive buffer of a collective the host framework ran). */
int fxjps_set_grid_device(fxjps_t* h, const void* d_occ, int32_t W, int32_t H);

/* Callers' grid preparation on the device -- replaces
 * scripts/global_planner_st.py:230-272 (variant 0: dilation offsets {-ifa,0,ifa}^2, start/goal shifted by
 * map_d - 1) and scripts/global_planner_ccst.py:415-458 (variant 1: full (2*ifa+1)^2 dilation, shift map_d):
 * zero-pad `raw` (W0 x H0, non-zero = occupied) so that start and goal fit, dilate, make the result the
 * resident grid, shift start_xy / goal_xy (in: cell indices relative to `raw`, may be negative; out: indices
 * in the prepared grid) and move a goal that fell on an obstacle to the nearest free cell of its row, else of
 * its column.  out_map_d receives the low-side padding (dx, dy).  out_end_occu (may be NULL) receives the
 * reference's `end_occu` flag: variant 0 -- the shifted goal was on an obstacle (global_planner_st.py:268-275);
 * variant 1 -- any occupied cell in mapu[gx-ifa:gx+ifa, gy-ifa:gy+ifa] around the (moved) goal
 * (global_planner_ccst.py:461-464).  It is the `end_occu` argument of fxjps_waypoint_st / _ccst. */
int fxjps_prepare_grid(fxjps_t* h, const uint8_t* raw, int32_t W0, int32_t H0, int32_t ifa, int32_t variant,
                       int32_t* start_xy, int32_t* goal_xy, int32_t* out_W, int32_t* out_H, int32_t* out_map_d,
                       int32_t* out_end_occu);

/* Same, straight from a nav_msgs/OccupancyGrid: `data` is the message's int8 data[] (row-major [y][x],
 * width = x extent, height = y extent).  Fuses map_callback (global_planner_st.py:15-20,
 * global_planner_ccst.py:17-23: reshape(h, w).T, 100 -> 1, -1 -> 0) into the preparation kernel. */
int fxjps_prepare_occupancy_msg(fxjps_t* h, const int8_t* data, int32_t width, int32_t height, int32_t ifa,
                                int32_t variant, int32_t* start_xy, int32_t* goal_xy, int32_t* out_W,
                                int32_t* out_H, int32_t* out_map_d, int32_t* out_end_occu);

/* Copy the resident grid back (out may be NULL to query the size only). */
int fxjps_get_grid(fxjps_t* h, uint8_t* out, int32_t* out_W, int32_t* out_H);
/* The same for context `ctx` of a multi-device handle (0 .. contexts - 1, fxjps_comm_info): the bytes THAT device holds
 * after the broadcast of fxjps_set_grid -- SURVEY.md 4 T4 asks for the grid hash to be equal on every device. */
int fxjps_get_grid_context(fxjps_t* h, int32_t ctx, uint8_t* out, int32_t* out_W, int32_t* out_H);

/* ---- Wire / on-disk adapters; device-side byte transposes of the resident grid.
 *
 * fxjps_publish_map: the inverse of map_callback, what publish_map (scripts/global_planner_st.py:102-115) puts into
 * the nav_msgs/OccupancyGrid it publishes: info.width = W (len(data)), info.height = H (len(data[0])),
 * data[y*W + x] = 100 where the grid is occupied, else 0 (`data.T.reshape(...)`).  out_data holds W*H int8 (NULL:
 * sizes only).
 *
 * fxjps_set_grid_image: the prior-map loader convention of scripts/global_planner_st.py:176-182 on the decoded 8-bit
 * grey image (`img.convert('L')`, rows x cols, row-major): pixel > 200 is free, anything else occupied, and
 * map_pre = img[::-1].T, i.e. W = cols, H = rows, grid[x][y] = pixel[rows-1-y][x].  The result becomes the resident
 * grid (like fxjps_set_grid).
 *
 * fxjps_snapshot_image: the snapshot convention of scripts/global_planner_st.py:365-374 (`mapsave.T[::-1]`): an
 * H-row x W-column image, pixel[r][x] = 255 where grid[x][H-1-r] is free, else 0; channels = 1 ('L') or 3 (the
 * `.convert('RGB')` replication).  out holds H*W*channels bytes (NULL: sizes only).  set_grid_image(snapshot_image)
 * reproduces the grid. */
int fxjps_publish_map(fxjps_t* h, int8_t* out_data, int32_t* out_width, int32_t* out_height);
int fxjps_set_grid_image(fxjps_t* h, const uint8_t* gray, int32_t rows, int32_t cols);
int fxjps_snapshot_image(fxjps_t* h, uint8_t* out, int32_t channels, int32_t* out_rows, int32_t* out_cols);

/* Streaming replan: set n cells (xy pairs) to val[i] (0 free / non-zero obstacle) on the resident grid and rebuild the
 * derived maps -- only what the changed cells can reach (their box for the scan words, the rows and columns through it
 * for the cell infos; small lists are united into the component labels).  Cells outside the grid are ignored; a cell
 * named more than once takes the value of its last entry. */
int fxjps_update_cells(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n);

/* fxjps_update_cells without the rebuild of the derived maps: the updates are applied to the resident grid (calls are
 * applied in order; a cell named more than once in a call takes its last entry; xy / val are copied before the call returns, which it may do
 * while the update is still queued on the device -- every reader of the grid, fxjps_get_grid included, is ordered
 * behind it) and the maps are rebuilt once by the next fxjps_update_cells, fxjps_plan_batch* or fxjps_replan_frame.  For hosts that hand a handle several frames' updates before it plans
 * again (fuxi_planner_amd.replan.FramePipeline; scripts/global_planner_st.py:15-25 delivers one map message per
 * callback, the node plans once per tick). */
int fxjps_update_cells_deferred(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n);

/* Streaming replan with a persistent query set (SURVEY.md 8f, row N4 / BASELINE config 5: the node replans the same
 * goals tick after tick, scripts/global_planner_ccst.py:476-480).  fxjps_set_queries stores nq (start, goal) pairs,
 * hchoice and max_path_len in the handle (copied).  fxjps_replan_frame applies one frame of cell updates (as
 * fxjps_update_cells; n may be 0), rebuilds the derived maps and plans the stored queries against the new grid, all
 * queued on the device without an intermediate host wait; results as fxjps_plan_batch_csr (fxjps_last_cells and
 * fxjps_last_timing work the same way).  Results are bit-identical to set_grid + plan on the updated grid.  Exact
 * reuse: every search records its read set (which of at most 64 x 64 grid tiles hold a cell whose derived data it
 * read); a stored result whose read set no updated cell, nor any of its 8 neighbours, falls into is what a
 * from-scratch search would return and is handed back without searching (fxjps_timing_t.reused counts them).  Any
 * other call that changes the grid or the device result buffers (set_grid*, update_cells, plan_batch*) drops the
 * stored results; FXJPS_REPLAN_REUSE=0 in the environment turns the reuse off. */
int fxjps_set_queries(fxjps_t* h, const int32_t* starts_xy, const int32_t* goals_xy, int64_t nq, int32_t hchoice,
                      int32_t max_path_len);
int fxjps_replan_frame(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n, int64_t* out_offsets,
                       int32_t* out_cells_xy, int64_t cells_capacity, int32_t* out_len, double* out_cost,
                       double* out_seconds_total);

/* Plan nq independent (start, goal) queries against the resident grid.
 *   starts_xy, goals_xy : nq (x, y) pairs
 *   out_cells_xy        : nq * max_path_len (x, y) pairs; query q's jump points
 *                         start at out_cells_xy + q*max_path_len*2 (may be NULL)
 *   out_len             : nq; >0 number of jump points, else a FXJPS_Q_* code
 *   out_cost            : nq float64 path costs (gscore[goal], jps1.py:207); 0 if no path
 *   out_seconds_total   : wall seconds spent inside the call (may be NULL)
 */
int fxjps_plan_batch(fxjps_t* h, const int32_t* starts_xy, const int32_t* goals_xy, int64_t nq,
                     int32_t hchoice, int32_t max_path_len, int32_t* out_cells_xy,
                     int32_t* out_len, double* out_cost, double* out_seconds_total);

/* Same search, compact result: out_offsets has nq+1 entries, query q's jump
 * points are out_cells_xy[2*out_offsets[q] .. 2*out_offsets[q+1]).  Queries
 * without a path contribute zero cells.  cells_capacity is the number of
 * (x, y) pairs out_cells_xy can hold.  Pass out_cells_xy == NULL to get
 * out_len/out_cost/out_offsets only; the cells of the batch stay in the
 * handle and fxjps_last_cells() copies them once the caller has sized a
 * buffer from out_offsets[nq].  A non-NULL buffer that is too small makes
 * the call return FXJPS_E_ARG (everything but the cells is filled in). */
int fxjps_plan_batch_csr(fxjps_t* h, const int32_t* starts_xy, const int32_t* goals_xy,
                         int64_t nq, int32_t hchoice, int32_t max_path_len,
                         int64_t* out_offsets, int32_t* out_cells_xy, int64_t cells_capacity,
                         int32_t* out_len, double* out_cost, double* out_seconds_total);

/* Copy the jump points of the most recent batch (CSR order) into out_cells_xy. */
int fxjps_last_cells(fxjps_t* h, int32_t* out_cells_xy, int64_t cells_capacity);

/* Measurement hooks (bench.py, tests). */
typedef struct fxjps_timing {
    double search_kernel_ms; /* HIP-event time of the search kernel launches of the last batch */
    double total_ms;         /* wall time of the last batch call */
    int64_t search_launches; /* kernel launches that made up search_kernel_ms (a batch of 4 096 .. 32 768 queries is two
                                overlapping launches: its longest queries on CUs of their own, the rest beside them) */
    int64_t retried;         /* queries re-run with large scratch */
    int64_t pops;            /* open-list pops executed by the last batch (all devices) */
    int64_t pushes;
    int64_t far_refills;     /* open-list refills from the global-memory tier */
    int64_t slow_pops;       /* pops taken straight from the global-memory tier (> 256 entries tied at the minimum key) */
    int64_t table_wipes;     /* visited-table wipes after a wavefront's generation counter wrapped (every 63 searches) */
    int64_t reused;          /* fxjps_replan_frame: stored results returned without a search (their read set was untouched) */
    int64_t table_direct;    /* 1: the last batch ran on visited tables indexed by the cell (grids of up to 2^20 slots), 0: on
                                hashed tables of 4-slot buckets (larger grids; FXJPS_DIRECT=0) */
    int64_t waves;           /* resident wavefronts (= queries in flight) the last batch ran with, summed over the contexts */
    int64_t waves_short;     /* 1: a scratch pool was granted fewer wavefronts than the batch asked for (memory budget of the
                                handle, or the device ran out of memory): the batch ran, with less parallelism */
    double head_launch_ms;   /* search_launches == 2: HIP-event time of the launch of the longest queries alone ... */
    double batch_launch_ms;  /* ... and of the launch of the rest of the batch beside it (0 when the batch was one launch) */
    int64_t solo_timeouts;   /* waits for the head launch's blocks to report from their CUs that ran into their 5 ms bound,
                                since the handle was created (3 in a row on a device: it runs its batches as one launch from
                                then on; a lone one is the cold first launch of a kernel) */
} fxjps_timing_t;
/* fxjps_last_timing writes sizeof(fxjps_timing_t) bytes AS THE LIBRARY WAS BUILT: a host compiled against an older header
 * checks fxjps_version() == FXJPS_VERSION (or fxjps_timing_size() == sizeof(fxjps_timing_t)) first, or calls
 * fxjps_last_timing_sized, which writes at most out_size bytes (the struct only grows at its end). */
int fxjps_last_timing(fxjps_t* h, fxjps_timing_t* out);
int fxjps_timing_size(void);
int fxjps_last_timing_sized(fxjps_t* h, void* out, int64_t out_size);

/* Per-context figures of the last batch (ctx = 0 .. contexts-1, the order of device_ids at fxjps_create): the device
 * it ran on, the queries of its contiguous shard, the HIP-event time of its search kernel launches, its resident
 * wavefronts.  Any out pointer may be NULL. */
int fxjps_last_timing_device(fxjps_t* h, int32_t ctx, int32_t* out_device, int64_t* out_nq, double* out_kernel_ms,
                             int64_t* out_waves);

/* What the handle spans: its contexts (entries of device_ids), the distinct devices among them, and the number of
 * ranks of its RCCL communicator (ncclCommCount; 0 while no collective has run: a single device, or contexts that
 * share one -- the grid then travels by device-to-device copies).  north_star: one ncclBroadcast of the grid over
 * xGMI per fxjps_set_grid, no other collective. */
int fxjps_comm_info(fxjps_t* h, int32_t* out_contexts, int32_t* out_devices, int32_t* out_rccl_ranks);

/* Several handles on one device (fuxi_planner_amd.replan.FramePipeline keeps K of them): each handle sizes its scratch
 * pools and its resident wavefronts as if it owned 1/handles_per_device of every device it spans.  Default 1. */
int fxjps_set_memory_share(fxjps_t* h, int32_t handles_per_device);

/* Device self-test: sqrt((double)n) for n in [n0, n1) written to out (host).
 * Used by the tests to prove the device square root is the correctly rounded
 * one math.sqrt gives (jps1.py:12,246). */
int fxjps_selftest_sqrt(fxjps_t* h, uint32_t n0, uint32_t n1, double* out);

/* Device self-test of the wavefront-wide DPP minimum used by the open list against the
 * shuffle form and a host reference, on `rounds` rows of 64 pseudo-random values. */
int fxjps_selftest_wavemin(fxjps_t* h, int32_t rounds, uint64_t seed, int64_t* mismatches);

/* Device self-test of the open list alone: one wavefront runs a script of nsteps steps through the open-list code of
 * the search kernel (registers / LDS / global-memory tiers; banded != 0: the far band as a ring of f bands) -- step i
 * pops up to step_pops[i] entries (as many as the register tier holds, at least one while the list is not empty) and
 * then pushes the keys [step_off[i], step_off[i+1]) (at most 64; key = (keys_f, keys_x): f as raw bits, packed
 * x:13|y:13|direction:4) -- and then pops until the list is empty.  out_f / out_x / out_slot (nkeys each) receive the
 * popped keys and the index of the pushed entry each one was, in pop order; out_k[i] the number of pops of step i;
 * out_info[0] the total popped, out_info[1] a failure code (0: none, 1 / 2: a far-tier region was full, 3: more pops
 * than pushes), out_info[2] the refills from the global-memory tier, out_info[3] the pops taken straight from it (more
 * than 256 entries with one and the same full key), out_info[4 .. 7] what the register / LDS / near / far tiers held when
 * the script ended; out_info holds 8 values.  far_cap / near_max size the global-memory tier as the planner's scratch configuration would (tests
 * make them tiny); delta0 is the first refill width.  The host checks the pops against a binary heap. */
int fxjps_selftest_openlist(fxjps_t* h, int32_t banded, int32_t far_cap, int32_t near_max, double delta0, const uint64_t* keys_f,
                            const uint32_t* keys_x, int64_t nkeys, const uint32_t* step_pops, const uint32_t* step_off, int32_t nsteps,
                            uint64_t* out_f, uint32_t* out_x, uint32_t* out_slot, uint32_t* out_k, uint32_t* out_info);

/* Copy the derived device maps back for inspection (tests): the padded
 * (W+2)x(H+2) neighbour-mask bytes.  buf must hold (W+2)*(H+2) bytes. */
int fxjps_debug_read_nbmask(fxjps_t* h, uint8_t* buf);

/* The derived device maps as they are (tests compare them after cell updates -- which rebuild only what the changed
 * cells can reach -- with those of a fresh upload): which = 0 the scan words ([4][LINES][WORDS] pairs of u64 {stop, occ},
 * LINES = max(W, H) + 2, WORDS = ceil(LINES / 64)), 1 the cell infos (u16 [W + 2][NS], NS = H + 2 rounded up to 64; the
 * columns from H + 2 on are unused), 2 the component forest (int32 [W][H]: parent links, a root points at itself, -1
 * never free), 3 the neighbour bytes ([W + 2][NS]), 4 the diagonal scan words ([4][W + H + 3][WORDS] pairs of u64 {stop, occ}:
 * travel directions (+,+), (-,-), (+,-), (-,+), one line per diagonal, bit = padded x), 5 the jump distances (u16
 * [W + 2][NS][8]: per cell and direction, steps to the cell that ends the goal-free jump | "it is a jump point" << 15;
 * directions in the order (-1,-1), (-1,0), (-1,1), (0,-1), (0,1), (1,-1), (1,0), (1,1)).  out_bytes receives the size;
 * buf == NULL: the size only. */
int fxjps_debug_read_maps(fxjps_t* h, int32_t which, void* buf, int64_t capacity_bytes, int64_t* out_bytes);

/* Measurement aids of tools/ (not used by the planner's Python host code).  fxjps_debug_counters: the 64 raw device
 * counters of the last batch on the first context ([0] pops, [1] pushes, [2] far refills, [3] slow pops, [7] table wipes;
 * the rest is filled by the diagnostic build -DFXJPS_PROF only).  fxjps_debug_qstat: with FXJPS_QSTAT=1 in the
 * environment, 4 u64 per query of the last batch on the first context: start, end (100 MHz ticks), pops, wavefront (low
 * 24 bits) | shader-clock cycles of the search << 24. */
int fxjps_debug_counters(fxjps_t* h, unsigned long long* out64);
int fxjps_debug_qstat(fxjps_t* h, unsigned long long* out, int64_t nq);

/* ---- Waypoint selection after a plan.  One path per call, host functions (no device work, no handle;
 * the batch forms below take the grid from the handle and run the ccst pruning on the device):
 * the step the reference's nodes run on the path jps1.method returned.  `cells` are the n (x, y) jump points of
 * one query as fxjps_plan_batch(_csr) returns them.
 *
 * fxjps_waypoint_st: scripts/global_planner_st.py:292-327 (angle / distance rule).  map_start is the shifted start
 * of the tick, prev_wp the waypoint left over from the previous tick (NULL: None; prev_dim 2 or 3 components) --
 * the reference keeps it when the loop does not pick a new one.  out_wp has out_dim (2 or 3) valid components,
 * out_goal is global_goal after the block (it becomes the vehicle position when end_occu == 1), out_ang_wp ang_wp.
 *
 * fxjps_waypoint_ccst: scripts/global_planner_ccst.py:487-544 with map_line_col (:258-283): points closer than
 * 1.5 to the vehicle are dropped, then every point whose neighbours see each other on the grid (occ, uint8 [W][H],
 * obstacle iff == 1: the matrix the search ran on); the waypoint is the 1.4 / 0.6 blend of the second and third
 * remaining points, or the goal; with end_occu == 1 (:541-544) the vehicle position becomes both waypoint and goal.
 * out_goal (3 doubles, optional) is global_goal after the block.  kept_cells (2 * n int32, optional) / n_kept
 * receive the remaining cells. */
int fxjps_waypoint_st(const int32_t* cells, int32_t n, const int32_t* map_start, double reso, const double* origin, const double* pos,
                      const double* goal, int32_t end_occu, double dis_wp_tre, double ang_wp_tre, const double* prev_wp,
                      int32_t prev_dim, double* out_wp, int32_t* out_dim, double* out_goal, double* out_ang_wp);
int fxjps_waypoint_ccst(const int32_t* cells, int32_t n, const uint8_t* occ, int32_t W, int32_t H, double reso, const double* origin,
                        const double* pos, const double* goal, int32_t end_occu, double* out_wp, double* out_goal,
                        int32_t* kept_cells, int32_t* n_kept);

/* ---- The same for every path of a batch.  offsets / cells_xy: the paths as fxjps_plan_batch_csr returns them (nq + 1
 * offsets, (x, y) pairs); both NULL: the paths of the handle's most recent batch, which are still resident on the
 * device(s) -- nq must be that batch's.  A query without a path gets the goal as its waypoint (`wp = global_goal`,
 * scripts/global_planner_st.py:287-290, scripts/global_planner_ccst.py:481-485).  pos, goal, out_wp, out_goal are nq x 3
 * doubles, end_occu nq flags (NULL: all 0), reso and origin one value for the batch (one grid).
 *
 * fxjps_waypoint_ccst_batch runs on the device: one wavefront per path against the RESIDENT grid (the matrix the search
 * ran on; nothing is passed again) -- near-point deletion, the line-of-sight pruning with map_line_col's float64
 * raster (np.arange / np.rint / astype(int) are IEEE division, multiplication and round-half-even on the device), the
 * 1.4 / 0.6 blend; results bit-identical to fxjps_waypoint_ccst path by path.  out_n_kept[q] receives the number of
 * remaining points, out_kept_cells (optional, kept_capacity pairs >= offsets[nq]) the remaining cells of path q at
 * offsets[q] (the offsets of the paths themselves).
 *
 * fxjps_waypoint_st_batch runs on the device too (version 600+): one wavefront per path, the rule's loop as a comparison
 * of neighbouring lanes.  Its decisions -- and the angle it returns -- hang on libm's atan2 of integer pairs
 * (cell + 1 - map_start), as CPython's math.atan2 does: the device looks them up in a table that this call fills with
 * the HOST's atan2 (nthreads host threads, 0: all cores) for the range of pairs the batch can ask for, once per range,
 * and keeps on the device (17 MB for a 1024 x 1024 grid).  A map_start so far off the grid that the table would exceed
 * 2^27 entries makes the call walk the batch on nthreads host threads instead.  map_start is nq x 2, prev_wp nq x 3
 * with prev_dim[q] in {0: None, 2, 3} (both NULL: no previous waypoints), out_dim / out_ang_wp nq values; results
 * bit-identical to fxjps_waypoint_st path by path. */
int fxjps_waypoint_ccst_batch(fxjps_t* h, int64_t nq, const int64_t* offsets, const int32_t* cells_xy, double reso, const double* origin,
                              const double* pos, const double* goal, const int32_t* end_occu, double* out_wp, double* out_goal,
                              int32_t* out_n_kept, int32_t* out_kept_cells, int64_t kept_capacity);
int fxjps_waypoint_st_batch(fxjps_t* h, int64_t nq, const int64_t* offsets, const int32_t* cells_xy, const int32_t* map_start, double reso,
                            const double* origin, const double* pos, const double* goal, const int32_t* end_occu, double dis_wp_tre,
                            double ang_wp_tre, const double* prev_wp, const int32_t* prev_dim, double* out_wp, int32_t* out_dim,
                            double* out_goal, double* out_ang_wp, int32_t nthreads);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif
