// fxjps.hip -- host side of libfxjps.so: the C ABI of include/fxjps.h on top of
// the HIP kernels in fxjps_kernels.hip.inc.  gfx950 only; no CPU fallback: if
// there is no HIP device every entry point fails with FXJPS_E_NODEV.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see Makefile).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fxjps.h"
#include "fxjps_kernels.hip.inc"

namespace {

using fx::FarEnt;
using fx::GridDev;
using fx::SearchArgs;
using fx::TEnt;

// (handles are created from threads: the text of a failed fxjps_create* belongs to the calling thread, which is the one that
// asks for it with fxjps_last_error(NULL))
thread_local std::string g_create_error;

// FXJPS_DEBUG=1 in the environment: progress lines on stderr (bring-up aid)
bool dbg_on() {
    static int on = -1;
    if (on < 0) on = getenv("FXJPS_DEBUG") ? 1 : 0;
    return on == 1;
}
#define DBG(...)                          \
    do {                                  \
        if (dbg_on()) {                   \
            fprintf(stderr, "[fxjps] ");  \
            fprintf(stderr, __VA_ARGS__); \
            fprintf(stderr, "\n");        \
            fflush(stderr);               \
        }                                 \
    } while (0)

// Independent jobs side by side on host threads (one per context of a multi-device handle; slices of one large copy).
// Nothing may leave the library as a C++ exception: a thread that cannot be created makes its job run on the calling
// thread instead, and a job that throws (std::bad_alloc in a worker's vectors) is reported as FXJPS_E_NOMEM.  -> the first
// non-zero result in job order.
template <typename F>
int run_side_by_side(size_t n, F&& job) {  // job(i) -> int
    std::vector<int> rcs(n, FXJPS_OK);
    const auto guarded = [&](size_t i) {
        try {
            rcs[i] = job(i);
        } catch (const std::bad_alloc&) {
            rcs[i] = FXJPS_E_NOMEM;
        } catch (...) {
            rcs[i] = FXJPS_E_HIP;
        }
    };
    std::vector<std::thread> th;
    try {
        th.reserve(n);
    } catch (...) {
    }
    for (size_t i = 1; i < n; i++) {
        bool started = false;
        try {
            th.emplace_back(guarded, i);
            started = true;
        } catch (...) {  // std::system_error: no more threads
        }
        if (!started) guarded(i);
    }
    if (n > 0) guarded(0);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < n; i++)
        if (rcs[i]) return rcs[i];
    return FXJPS_OK;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
struct DBuf {  // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct HBuf {  // grow-only pinned host buffer
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ScratchCfg {
    uint32_t nwaves = 0;
    uint32_t nbuckets = 0;    // buckets per wavefront table (hashed: any number; cell-indexed: slots / BUCKET)
    uint32_t usable = 0;      // hashed: entries a search may insert (3/4 of a full-size table, 7/8 of a shrunk one)
    uint32_t direct_ly = 0;  // > 0: the table is indexed by the cell (slot = x << direct_ly | y), see ensure_pool
    uint32_t far_cap = 0;
};

struct DevCtx {
    int dev = -1;
    int n_cu = 256;
    int share = 1;  // contexts of this handle on the same physical device (they split its memory and wavefront slots)
    size_t mem_total = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_solo = nullptr;  // the launch of the longest queries, one wavefront per CU, beside the batch's launch
    uint32_t* solo_started = nullptr;   // pinned host word: blocks of such launches that have started (only ever counts up)
    uint32_t solo_target = 0;           // ... and how many have been launched
    int solo_timeouts = 0;              // waits for that counter that ran into their 5 ms bound, since the handle was created
    int solo_timeouts_run = 0;          // ... in a row (3: no more head launches on this device; a cold first launch of a
                                        // kernel -- its code object is loaded then -- is a lone timeout and means nothing)
    int xcc_rr = -1;                    // 1: the blocks of a grid run on XCD b % 8 (probed once); 0: not so; -1: not probed yet
    size_t cells_bound = 0;             // bytes the batch in progress may still allocate for its packed paths
    bool lds_attr_done[8] = {};         // k_search instantiations whose dynamic-LDS limit has been raised on this device
    bool lds_attr_coop[2] = {};         // ... and k_search_coop's
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_solo0 = nullptr, ev_solo1 = nullptr;
    hipEvent_t ev_hd0 = nullptr, ev_hd1 = nullptr, ev_bt0 = nullptr, ev_bt1 = nullptr;  // around the head launch / the batch's launch alone
    bool had_solo = false;              // the last regular-pool launch was two launches
    double head_ms = 0, batch_ms = 0;   // ... and how long each of them ran
    // grid
    int W = 0, H = 0, PW = 0, PH = 0, NS = 0, LINES = 0, WORDS = 0, tsh = 0;
    DBuf<uint8_t> occ, nb8;
    DBuf<fx::BmWord> bm;
    DBuf<int> comp;
    DBuf<uint16_t> ci;
    DBuf<uint16_t> jd;  // jump distances [PW][NS][8] (k_derive_jd)
    // search scratch (two pools: the regular one and the large retry one)
    DBuf<TEnt> tables[2];
    DBuf<FarEnt> far[2];
    DBuf<uint32_t> wave_gen[2];
    ScratchCfg cfg[2];
    bool pool_clean[2] = {false, false};
    // batch buffers
    DBuf<int32_t> d_starts, d_goals, d_len, d_cells;
    DBuf<double> d_cost;
    DBuf<uint32_t> d_path, d_order, d_redo;
    DBuf<long long> d_offsets;
    DBuf<unsigned int> d_next;
    DBuf<unsigned long long> d_counters, d_qstat, d_qread;
    std::vector<unsigned long long> h_qread;  // read sets of the resident results (streaming replan)
    std::vector<uint8_t> h_sel;   // mode 2: which queries of the shard are searched again this frame
    int mode = 0;                 // 0 plain batch, 1 search everything and record read sets, 2 the same for h_sel only
    int64_t nrun = 0;             // queries handed to the search kernel
    DBuf<uint8_t> d_raw, d_img;
    // waypoint selection over a batch (fxjps_waypoint_ccst_batch)
    DBuf<double> d_wp_in, d_wp_out;
    DBuf<int32_t> d_wp_eo, d_wp_nkept, d_wp_kept, d_wp_cells, d_wp_len;
    DBuf<long long> d_wp_off;
    // ... the st rule (fxjps_waypoint_st_batch): per-query inputs / outputs, and the host libm's atan2 of integer pairs
    DBuf<int32_t> d_wp_ms, d_wp_pdim, d_wp_dim;
    DBuf<double> d_wp_prev, d_wp_ang, d_atab;
    int atab_a = -1, atab_b = -1;  // d_atab holds atan2(a, b) for a in [0, atab_a], b in [-atab_b, atab_b]
    DBuf<int32_t> d_upd_xy;
    DBuf<uint8_t> d_upd_chg;
    DBuf<int> d_owner;        // [W][H], -1 at rest: which entry of an update list decides a cell it names several times
    bool owner_ready = false;
    // a partial rebuild's changed cells (k_derive_cellinfo -> k_jd_walk): marks [PW][NS], zero at rest; their list; counters
    DBuf<uint8_t> d_chgmap;
    DBuf<uint32_t> d_chglist;
    DBuf<unsigned int> d_chgcnt;
    DBuf<uint32_t> d_chgovf;  // walks handed on to k_jd_finish: 2 words each
    bool chg_ready = false;
    // what the cell updates since the last rebuild of the derived maps can have changed: the box, in padded
    // coordinates, of the updated cells and their neighbours; whether the component labels need the full relabelling (a
    // large update, or 64 small ones, which are united into the existing labels as they come: k_ccl_update)
    bool dirty = false;
    int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
    bool ccl_full = false;
    int ccl_small = 0;
    // pinned staging of a cell-update list: the caller's arrays are copied here before the asynchronous H2D copy, so
    // that they need not outlive the call (ev_upd: the last copy out of the staging buffers has completed)
    HBuf<int32_t> h_upd_xy;
    hipEvent_t ev_upd = nullptr;
    HBuf<uint8_t> h_occ_stage;      // a small grid on its way to the device (fxjps_set_grid returns while it travels)
    hipEvent_t ev_stage = nullptr;  // ... the copy out of it has completed
    bool stage_pending = false;
    hipEvent_t ev_ccl0 = nullptr, ev_ccl1 = nullptr;  // a whole map build: the component labels on the second stream, beside the maps
    bool upd_pending = false;
    HBuf<int32_t> h_len, h_cells;
    HBuf<uint32_t> h_path1;       // single calls: the search kernel writes the packed path of its one query here
    HBuf<double> h_cost;
    HBuf<long long> h_offsets;
    HBuf<unsigned long long> h_counters;
    std::vector<uint32_t> h_order;
    // shard of the current batch
    int64_t q0 = 0, nq = 0;
    double kernel_ms = 0;
    int wall_khz = 100000;  // hipDeviceAttributeWallClockRate (wall_clock64 ticks per ms)
    int64_t launches = 0, retried = 0;
    uint32_t waves_used = 0;  // resident wavefronts of the last regular-pool launch
    bool waves_short = false; // a scratch pool was granted fewer wavefronts than the batch asked for (memory budget / out of memory)
};

}  // namespace

struct fxjps {
    std::vector<DevCtx> devs;
    std::string err;
    std::mutex err_mu;  // the shards of a multi-device batch run on one host thread each: whichever fails last leaves its text
    bool have_grid = false;
    bool maps_stale = false;  // cell updates were applied without rebuilding the derived maps (fxjps_update_cells_deferred)
    int mem_div = 1;          // handles sharing each device (fxjps_set_memory_share): the scratch budgets are divided by it
    fxjps_timing_t timing{};
    int64_t last_nq = 0;
    bool last_on_host = false;  // the last batch was a single call: its CSR is in the pinned host buffers only
    // persistent query set of the streaming-replan entry points (fxjps_set_queries / fxjps_replan_frame)
    std::vector<int32_t> q_starts, q_goals;
    int q_hchoice = 0, q_max_len = 0;
    bool q_set = false;
    // true while the device result buffers and the host read sets of every device describe the stored queries on the
    // resident grid: then a frame only searches the queries whose read set its cell updates touch
    bool q_results_valid = false;
    // RCCL (only for n_dev > 1), resolved with dlopen so that a single-GPU
    // deployment does not need librccl at load time
    void* rccl = nullptr;
    std::vector<void*> comms;
    // one process per GPU without torch (fxjps_create_rank): this handle is rank `rank` of `world`, its one communicator
    // (comms[0]) was made with ncclCommInitRank from the id rank 0 handed out
    int rank = -1, world = 0;
};

namespace {

int fail(fxjps* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) {
        std::lock_guard<std::mutex> lk(h->err_mu);
        h->err = buf;
    } else {
        g_create_error = buf;
    }
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail(h, e__ == hipErrorOutOfMemory ? FXJPS_E_NOMEM : FXJPS_E_HIP, "%s: %s (%s:%d)", \
                        #call, hipGetErrorString(e__), __FILE__, __LINE__);                      \
    } while (0)

// nodeNeighbours (jps1.py:49-93) on a neighbour mask: which direction each of the 8 rays of a node
// follows in the search kernel.  pd = (dX+1)*4 + (dY+1) of direction(c, came_from[c]) (jps1.py:40-47),
// 5 = no parent (the start node, `type(parent) != tuple`, :51).
uint32_t dirlut_entry(uint32_t pd, uint32_t nbm) {
    auto occ = [&](int dx, int dy) -> bool { return (nbm >> fx::nbit(dx, dy)) & 1u; };
    auto blocked = [&](int dx, int dy) -> bool {  // blocked(c, dx, dy), jps1.py:14-31, border == occupied
        if (dx != 0 && dy != 0) return (occ(dx, 0) && occ(0, dy)) || occ(dx, dy);
        return occ(dx, dy);
    };
    uint32_t codes[8];
    int n = 0;
    auto add = [&](int dx, int dy) { codes[n++] = (uint32_t)((dx + 1) * 4 + (dy + 1)); };
    const int pdx = (int)(pd >> 2) - 1, pdy = (int)(pd & 3u) - 1;
    if (pd == fx::PD_NONE) {  // :51-56
        static const int all8[8][2] = {{-1, 0}, {0, -1}, {1, 0}, {0, 1}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
        for (auto& d : all8)
            if (!blocked(d[0], d[1])) add(d[0], d[1]);
    } else if (pdx < -1 || pdx > 1 || pdy < -1 || pdy > 1) {
        // not a direction code
    } else if (pdx != 0 && pdy != 0) {  // :59-73
        if (!blocked(0, pdy)) add(0, pdy);
        if (!blocked(pdx, 0)) add(pdx, 0);
        if ((!blocked(0, pdy) || !blocked(pdx, 0)) && !blocked(pdx, pdy)) add(pdx, pdy);
        if (blocked(-pdx, 0) && !blocked(0, pdy)) add(-pdx, pdy);
        if (blocked(0, -pdy) && !blocked(pdx, 0)) add(pdx, -pdy);
    } else if (pdx == 0) {  // :76-83; the guard at :77 tests the node itself, which is free
        if (!blocked(0, pdy)) add(0, pdy);
        if (blocked(1, 0)) add(1, pdy);
        if (blocked(-1, 0)) add(-1, pdy);
    } else {  // :85-92
        if (!blocked(pdx, 0)) {
            add(pdx, 0);
            if (blocked(0, 1)) add(pdx, 1);
            if (blocked(0, -1)) add(pdx, -1);
        }
    }
    uint32_t v = 0;
    for (int g = 0; g < 8; g++) v |= (g < n ? codes[g] : fx::DIR_NONE) << (4 * g);
    return v;
}

GridDev grid_of(const DevCtx& d) {
    GridDev G;
    G.bm = d.bm.p;
    G.ci = d.ci.p;
    G.nb8 = d.nb8.p;
    G.jd = d.jd.p;
    G.comp = d.comp.p;
    G.W = d.W;
    G.H = d.H;
    G.PW = d.PW;
    G.PH = d.PH;
    G.NS = d.NS;
    G.LINES = d.LINES;
    G.WORDS = d.WORDS;
    G.DLINES = d.PW + d.PH - 1;
    G.tsh = d.tsh;
    return G;
}

// (re)build nb8, the scan bitmaps, the cell infos and the component labels from d.occ.  whole == false: only what the
// cell updates since the last rebuild can have changed (DevCtx::dirty and its box) -- the neighbour bytes and scan
// words of the box, the cell infos of the rows and columns through it; the labels were kept up to date by
// k_ccl_update unless a full relabelling is due.
int derive_maps(fxjps* h, DevCtx& d, bool whole = true) {
    HIPCHK(h, hipSetDevice(d.dev));
    if (!whole && !d.dirty && !d.ccl_full) return FXJPS_OK;
    const bool box = !whole && d.dirty && (d.bx1 - d.bx0 + 1) * 2 <= d.PW && (d.by1 - d.by0 + 1) * 2 <= d.PH;
    const GridDev G = grid_of(d);
    // A whole build (a new grid): the component labels need nothing but the occupancy bytes, the five map kernels nothing of
    // the labels -- the three label kernels run on the handle's second stream beside them and join in front of whatever is
    // queued next (round 6: 27 of the 76 us of a 256 x 256 build; 221 of 382 us at 1024 x 1024).
    // A small grid: four launches instead of eight (k_build_1 .. 3, then k_derive_jd): what a build of tiny kernels waits for
    // is the runtime launching them one by one.  FXJPS_FUSED_BUILD=0: the separate kernels (test / measurement aid).
    const char* fused_env = getenv("FXJPS_FUSED_BUILD");  // (read per build: the tests compare the two forms in one process)
    if (whole && !(fused_env && atoi(fused_env) == 0) && (long long)d.W * d.H <= (1ll << 18)) {
        const long long ncell = (long long)d.W * d.H, npad = (long long)d.PW * d.PH;
        const unsigned nb_rows = (unsigned)(((long long)d.PW * d.WORDS + 3) / 4), nb_cols = (unsigned)(((long long)d.PH * d.WORDS + 3) / 4);
        const unsigned nb_cell = (unsigned)((ncell + 255) / 256), nb_ci = (unsigned)((npad + 255) / 256);
        const unsigned nb_diag = (unsigned)((4ll * G.DLINES * d.WORDS + 15) / 16), nb_flat = (unsigned)((ncell + 1023) / 1024);
        hipLaunchKernelGGL(fx::k_build_1, dim3(nb_rows + nb_cols + nb_cell), dim3(256), 0, d.stream, d.occ.p, G, d.nb8.p, d.bm.p, d.comp.p, nb_rows, nb_cols);
        hipLaunchKernelGGL(fx::k_build_2, dim3(nb_ci + nb_cell), dim3(256), 0, d.stream, d.occ.p, G, d.ci.p, d.comp.p, nb_ci);
        hipLaunchKernelGGL(fx::k_build_3, dim3(nb_diag + nb_flat), dim3(1024), 0, d.stream, d.occ.p, G, d.bm.p + (size_t)4 * d.LINES * d.WORDS, d.comp.p, nb_diag);
        hipLaunchKernelGGL(fx::k_derive_jd, dim3((unsigned)((npad * 8 + 255) / 256)), dim3(256), 0, d.stream, G, d.jd.p, fx::DiagRange{1, d.bx0, d.bx1, d.by0, d.by1});
        HIPCHK(h, hipGetLastError());
        d.ccl_small = 0;
        d.dirty = false;
        d.ccl_full = false;
        return FXJPS_OK;
    }
    const bool ccl_beside = whole && d.stream_solo != nullptr && d.ev_ccl0 != nullptr;
    const auto launch_ccl = [&](hipStream_t st) {
        const long long n = (long long)d.W * d.H;
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(fx::k_ccl_init, dim3(nb), dim3(256), 0, st, d.occ.p, n, d.H, d.comp.p);
        hipLaunchKernelGGL(fx::k_ccl_merge, dim3(nb), dim3(256), 0, st, d.occ.p, d.W, d.H, d.comp.p);
        hipLaunchKernelGGL(fx::k_ccl_flatten, dim3(nb), dim3(256), 0, st, n, d.comp.p);
    };
    if (ccl_beside) {
        HIPCHK(h, hipEventRecord(d.ev_ccl0, d.stream));  // (behind the copy / broadcast that brought the grid)
        HIPCHK(h, hipStreamWaitEvent(d.stream_solo, d.ev_ccl0, 0));
        launch_ccl(d.stream_solo);
        HIPCHK(h, hipEventRecord(d.ev_ccl1, d.stream_solo));
    }
    if (whole || d.dirty) {
        fx::MapRange rr{0, d.PW - 1, 0, d.WORDS - 1}, rc{0, d.PH - 1, 0, d.WORDS - 1};
        if (box) {
            rr = fx::MapRange{d.bx0, d.bx1, d.by0 >> 6, d.by1 >> 6};
            rc = fx::MapRange{d.by0, d.by1, d.bx0 >> 6, d.bx1 >> 6};
        }
        const long long nrw = (long long)(rr.l1 - rr.l0 + 1) * (rr.w1 - rr.w0 + 1), ncw = (long long)(rc.l1 - rc.l0 + 1) * (rc.w1 - rc.w0 + 1);
        hipLaunchKernelGGL(fx::k_derive_rows, dim3((unsigned)((nrw + 3) / 4)), dim3(256), 0, d.stream, d.occ.p, G, d.nb8.p, d.bm.p, rr);
        hipLaunchKernelGGL(fx::k_derive_cols, dim3((unsigned)((ncw + 3) / 4)), dim3(256), 0, d.stream, d.occ.p, G, d.bm.p, rc);
        fx::ChangeOut chg{nullptr, nullptr, nullptr, 0u, fx::MapRange{1, 0, 1, 0}};
        if (box) {  // the rows and the columns through the box (a straight jump ends where the line's next stop bit is)
            const fx::MapRange sa{d.bx0, d.bx1, 0, d.PH - 1}, sb{0, d.PW - 1, d.by0, d.by1};
            const long long na = (long long)(sa.l1 - sa.l0 + 1) * d.PH, nb = (long long)d.PW * (sb.w1 - sb.w0 + 1);
            // ... and which of those cells changed: where the re-scan of the jump distances starts (k_jd_walk)
            const bool walk = !(getenv("FXJPS_JD_WALK") && atoi(getenv("FXJPS_JD_WALK")) == 0);  // (0: measurement / test aid -- every record is read)
            if (walk) {
                // The change list is an optimisation with buffers of its own (up to 4 bytes per cell of the cross: 67 MB for
                // a half-map box at 4096^2): when the device has no room for them -- pool 0 of a lone handle may hold 80 % of
                // it -- the update does not fail, it takes round 4's way (every word of the diagonals through the cross
                // rebuilt, every record read: chg.map stays nullptr).
                bool room = true;
                if (!d.chg_ready) {  // (streaming callers only: with the first partial rebuild on a grid)
                    room = d.d_chgmap.ensure((size_t)d.PW * d.NS) == hipSuccess && d.d_chgcnt.ensure(4) == hipSuccess;
                    if (room) {
                        HIPCHK(h, hipMemsetAsync(d.d_chgmap.p, 0, (size_t)d.PW * d.NS, d.stream));
                        HIPCHK(h, hipMemsetAsync(d.d_chgcnt.p, 0, 4 * sizeof(unsigned int), d.stream));
                        d.chg_ready = true;
                    }
                }
                room = room && d.d_chglist.ensure((size_t)(na + nb)) == hipSuccess;
                constexpr uint32_t OVF_CAP = 16384;  // long walks handed on per update (2 words each); beyond it the records are streamed
                room = room && d.d_chgovf.ensure((size_t)2 * OVF_CAP) == hipSuccess;
                if (room) {
                    chg = fx::ChangeOut{d.d_chgmap.p, d.d_chglist.p, d.d_chgcnt.p, (uint32_t)(na + nb), fx::MapRange{d.bx0, d.bx1, d.by0, d.by1}, d.d_chgovf.p, OVF_CAP};
                    if (const char* e = getenv("FXJPS_JD_OVF_CAP")) chg.ovf_cap = (uint32_t)std::min(std::max(atoi(e), 0), (int)OVF_CAP);  // (test aid)
                } else
                    (void)hipGetLastError();  // (the failed allocation's sticky error: not this update's business)
            }
            hipLaunchKernelGGL(fx::k_derive_cellinfo, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, d.stream, G, d.ci.p, sa, sb, chg);
        } else {
            const long long ncell = (long long)d.PW * d.PH;
            hipLaunchKernelGGL(fx::k_derive_cellinfo, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, d.stream, G, d.ci.p,
                               fx::MapRange{0, d.PW - 1, 0, d.PH - 1}, fx::MapRange{1, 0, 1, 0}, chg);
        }
        {  // the diagonal scan words: a function of the cell infos (and the occupancy) cell by cell -- of the cells above
            const fx::DiagRange dr{box ? 0 : 1, d.bx0, d.bx1, d.by0, d.by1};
            const long long per = box ? (long long)((d.bx1 >> 6) - (d.bx0 >> 6) + 1) + (((d.by1 - d.by0 + 63) >> 6) + 1) : (long long)d.WORDS;
            const long long nw = 4ll * G.DLINES * per;
            if (box && chg.map != nullptr)  // (the changed cells alone, bit by bit)
                hipLaunchKernelGGL(fx::k_diag_update, dim3((unsigned)std::min<long long>(((long long)chg.cap * 4 + 255) / 256, 256)), dim3(256), 0, d.stream,
                                   d.occ.p, G, d.bm.p + (size_t)4 * d.LINES * d.WORDS, chg);
            else
                hipLaunchKernelGGL(fx::k_derive_diag, dim3((unsigned)((nw + 15) / 16)), dim3(1024), 0, d.stream, d.occ.p, G,
                                   d.bm.p + (size_t)4 * d.LINES * d.WORDS, dr);
            // ... and the jump distances: the goal-free jumps themselves, from every cell along every direction, read off
            // the scan words above (after an update: the entries whose old ray passes what the update can have changed)
            if (box && chg.map != nullptr) {
                // what the update can have changed, found from the changed cells backwards (a walker per changed cell and
                // direction) instead of by reading every record
                const int walk_max = getenv("FXJPS_JD_WALK_MAX") ? std::max(1, atoi(getenv("FXJPS_JD_WALK_MAX"))) : FXJPS_JD_WALK_MAX;  // (test aid)
                const long long nt = (long long)chg.cap * 8, nc = (long long)d.W * d.H;
                // (walks pay while the changed cells are few against the table: past W * H / 128 of them the records are streamed
                // -- measured at 4096^2 over obstacle densities 0 ... 0.2, profiles/r06_map_build.txt)
                const int div = getenv("FXJPS_JD_STREAM_DIV") ? std::max(1, atoi(getenv("FXJPS_JD_STREAM_DIV"))) : 128;  // (measurement / test aid)
                const uint32_t stream_over = (uint32_t)std::max<long long>(nc / div, 1024);
                hipLaunchKernelGGL(fx::k_jd_walk, dim3((unsigned)std::min<long long>((nt + 255) / 256, 2048)), dim3(256), 0, d.stream, G, d.jd.p, chg, dr, walk_max, stream_over);
                hipLaunchKernelGGL(fx::k_jd_finish, dim3((unsigned)std::min<long long>((nc + 255) / 256, 1024)), dim3(256), 0, d.stream, G, d.jd.p, chg, dr);
            } else if (box) {
                const long long nc = (long long)d.W * d.H;
                hipLaunchKernelGGL(fx::k_update_jd, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, d.stream, G, d.jd.p, dr);
            } else {
                const long long nj = (long long)d.PW * d.PH * 8;
                hipLaunchKernelGGL(fx::k_derive_jd, dim3((unsigned)((nj + 255) / 256)), dim3(256), 0, d.stream, G, d.jd.p, dr);
            }
        }
        HIPCHK(h, hipGetLastError());
    }
    if (ccl_beside) {
        HIPCHK(h, hipStreamWaitEvent(d.stream, d.ev_ccl1, 0));  // the join: whatever is queued on the main stream next sees the labels
        HIPCHK(h, hipGetLastError());
        d.ccl_small = 0;
    } else if (whole || d.ccl_full) {
        // component labels for the unreachable-goal early-out
        launch_ccl(d.stream);
        HIPCHK(h, hipGetLastError());
        d.ccl_small = 0;
    }
    d.dirty = false;
    d.ccl_full = false;
    return FXJPS_OK;
}

int alloc_grid(fxjps* h, DevCtx& d, int W, int H) {
    HIPCHK(h, hipSetDevice(d.dev));
    // The search scratch is sized by the grid's shape.  A new grid of the SAME shape keeps the pools as they are -- what
    // earlier searches left in the visited tables carries their generation tags and reads as empty, exactly as between two
    // batches on one grid -- so a caller that uploads a changed map every tick (jps1.method) does not pay for sizing and
    // wiping them again (~ 100 us of such a call); another shape makes the next batch size them anew.
    // (While one of the tests' sizing knobs is set every new grid makes the next batch size the pools anew, as before.)
    bool knob = false;
    for (const char* k : {"FXJPS_TABLE_LOG2", "FXJPS_TABLE_SHIFT", "FXJPS_FAR_CAP", "FXJPS_POOL_BUDGET_MB", "FXJPS_POOL_FRAC", "FXJPS_DIRECT",
                          "FXJPS_TABLE_SHRINK"})
        knob = knob || getenv(k) != nullptr;
    if (W != d.W || H != d.H || knob) {
        d.pool_clean[0] = d.pool_clean[1] = false;
        d.cfg[0] = ScratchCfg();
        d.cfg[1] = ScratchCfg();
        d.owner_ready = false;  // (the update lists' owner cells follow the grid's shape -- and are all -1 between two updates,
        d.chg_ready = false;    // as the marks of the changed cells are all zero: a grid of the same shape finds them in order)
    }
    d.W = W;
    d.H = H;
    d.PW = W + 2;
    d.PH = H + 2;
    d.NS = (d.PH + 63) & ~63;
    d.LINES = std::max(d.PW, d.PH);
    d.WORDS = (std::max(d.PW, d.PH) + 63) / 64;
    d.tsh = 0;  // read-set tiles (streaming replan): at most 64 x 64 of them cover the grid
    while (((std::max(W, H) - 1) >> d.tsh) > 63) d.tsh++;
    HIPCHK(h, d.occ.ensure((size_t)W * H));
    HIPCHK(h, d.comp.ensure((size_t)W * H));
    HIPCHK(h, d.nb8.ensure((size_t)d.PW * d.NS));
    HIPCHK(h, d.ci.ensure((size_t)d.PW * d.NS));
    HIPCHK(h, d.jd.ensure((size_t)d.PW * d.NS * 8));
    HIPCHK(h, d.bm.ensure((size_t)4 * d.LINES * d.WORDS + (size_t)4 * (d.PW + d.PH - 1) * d.WORDS));  // straight + diagonal scan lines
    return FXJPS_OK;
}

// wavefront counts are whole blocks of fx::WPB wavefronts (any block size: a measurement build runs ten per block)
constexpr uint64_t wpb_down(uint64_t v) { return v / (uint64_t)fx::WPB * (uint64_t)fx::WPB; }
constexpr uint64_t wpb_up(uint64_t v) { return wpb_down(v + (uint64_t)fx::WPB - 1u); }

uint32_t ceil_log2(uint64_t v) {
    uint32_t l = 0;
    while ((1ull << l) < v) l++;
    return l;
}

// Gives the memory of pool 0 back (a batch buffer did not fit beside it); the next batch sizes it again from what is free.
void release_pool0(DevCtx& d) {
    if (d.stream_solo) (void)hipStreamSynchronize(d.stream_solo);
    (void)hipStreamSynchronize(d.stream);
    d.tables[0].release();
    d.far[0].release();
    d.wave_gen[0].release();
    d.cfg[0] = ScratchCfg();
    d.pool_clean[0] = false;
}

// Size the per-wavefront scratch.  pool 0: many wavefronts, tables sized for the
// typical query; pool 1: few wavefronts, tables that cannot overflow.
int ensure_pool(fxjps* h, DevCtx& d, int pool, uint32_t want_waves) {
    const uint64_t cells = (uint64_t)d.W * d.H;
    ScratchCfg c;
    // Memory budget.  A handle that has the device to itself gives pool 0 up to 80 % of it, less what the batch still
    // has to allocate behind the search (the packed paths: at most nq * max_len cells), the smallest retry pool and a
    // margin -- on grids with hashed tables the number of resident wavefronts is what this budget buys, and the plans/s
    // of config 3 follow it almost linearly (1 536 / 2 270 / 2 840 wavefronts: 5.2 k / 7.5 k / 8.7 k plans/s).  Handles
    // that share a device keep to 60 % of their share.  Pool 1 (allocated while pool 0 stays resident) is sized from
    // what is free right now plus what it already holds, so that the two pools share one budget.  A batch buffer that
    // does not fit later takes the memory back (release_pool0 in run_shard).
    const size_t div = (size_t)(d.share * h->mem_div);
    double frac = div == 1 ? 0.8 : 0.6;
    if (const char* e = getenv("FXJPS_POOL_FRAC")) frac = std::min(0.95, std::max(0.05, atof(e)));  // measurement aid
    size_t budget = (d.mem_total ? (size_t)(d.mem_total * frac) : ((size_t)64 << 30)) / div;
    size_t free_b = 0, total_b = 0;
    const bool have_info = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    if (pool == 0 && div == 1 && have_info) {
        const size_t held = d.tables[0].cap * sizeof(TEnt) + d.far[0].cap * sizeof(FarEnt);
        const uint32_t l2r = std::max(ceil_log2(cells * 2 + 64), 12u);
        const size_t retry_wave = ((size_t)1 << l2r) * sizeof(TEnt) + (size_t)((cells * 2 + 1024) * 9 / 8) * sizeof(FarEnt);
        const size_t reserve = (size_t)fx::WPB * retry_wave + d.cells_bound + ((size_t)6 << 30);
        const size_t avail = free_b + held;
        budget = std::min(budget, avail > reserve ? avail - reserve : (size_t)0);
        budget = std::max(budget, std::min((size_t)(d.mem_total * 0.25), avail / 2));  // (never below what a shared device would get)
    }
    if (pool == 1 && have_info) {
        const size_t held = d.tables[1].cap * sizeof(TEnt) + d.far[1].cap * sizeof(FarEnt);
        budget = std::min(budget, (size_t)((free_b + held) * 0.9));
    }
    if (const char* e = getenv("FXJPS_POOL_BUDGET_MB")) budget = std::min<size_t>(budget, (size_t)std::max(1, atoi(e)) << 20);  // test aid
    if (pool == 0) {
        // measured on the config-2 workload (1024^2, 20 %): a reachable query inserts 34 k nodes
        // on average, 117 k at p99, 139 k at most, and keeps at most 2.4 k entries open; a table
        // of cells/4 entries (3/4 usable) covers all of them, the retry pool covers the rest
        uint32_t l2e = ceil_log2(std::max<uint64_t>(cells / 4, 1));  // entries
        // ... and up to four times that when a full set of wavefronts still fits in 40 % of the device: the hardest
        // queries fill half of a cells/4 table, where one 4-slot bucket in seven is full and the node that meets it
        // is committed by the slow general form (config 2: 152.9 -> 146.9 ms per 10 000 queries).
        {
            const uint64_t full = (uint64_t)d.n_cu * 4u * (uint64_t)fx::OCC / (uint64_t)(d.share * h->mem_div);
            const uint64_t cap = (d.mem_total ? (uint64_t)(d.mem_total * 0.4) : ((uint64_t)32 << 30)) / (uint64_t)(d.share * h->mem_div);
            uint32_t shift = 2;
            if (const char* e = getenv("FXJPS_TABLE_SHIFT")) shift = (uint32_t)std::max(0, atoi(e));  // measurement aid
            while (shift > 0 && full * (((uint64_t)19 << (l2e + shift))) > cap) shift--;  // 16-byte entries + the far tier (an eighth as many 18-byte entries)
            l2e += shift;
        }
        l2e = std::min(std::max(l2e, 12u), 23u);
        // A table with one slot per cell (slot = x << ly | y, both extents rounded up to powers of two) needs no
        // hashing, no buckets and cannot fill up: a probe is one 16-byte entry instead of a 64-byte bucket searched for
        // the cell.  Taken whenever a full set of wavefronts fits the same 40 % of the device with it (up to 2^20
        // slots: 1024^2, where it is exactly as large as the hashed table was); larger grids keep the hashed table.
        // FXJPS_DIRECT=0: measurement / test aid.
        {
            const uint32_t lx = std::max(ceil_log2((uint64_t)d.W), 1u), ly = std::max(ceil_log2((uint64_t)d.H), 1u);
            const uint32_t ld = std::max(lx + ly, 12u);
            const uint64_t full = (uint64_t)d.n_cu * 4u * (uint64_t)fx::OCC / (uint64_t)(d.share * h->mem_div);
            const uint64_t cap = (d.mem_total ? (uint64_t)(d.mem_total * 0.4) : ((uint64_t)32 << 30)) / (uint64_t)(d.share * h->mem_div);
            const bool allow = !(getenv("FXJPS_DIRECT") && atoi(getenv("FXJPS_DIRECT")) == 0) && !getenv("FXJPS_TABLE_LOG2");
            if (allow && ld <= 23u && full * ((uint64_t)19 << ld) <= cap) {
                l2e = ld;
                c.direct_ly = ly;
            }
        }
        // test aids: force a small table / far tier so that the overflow -> large-pool retry paths run
        if (const char* e = getenv("FXJPS_TABLE_LOG2")) l2e = (uint32_t)std::min(std::max(atoi(e), 4), 23);
        uint64_t entries = (uint64_t)1 << l2e;
        c.usable = (uint32_t)(entries / 4 * 3);
        // Hashed tables, and the wavefronts asked for do not fit the budget with them: the table shrinks, down to 3/4 of
        // its size (then filled up to 7/8 instead of 3/4: 88 % of the entries it could hold before), as far as it takes
        // to fit them.  The bucket count need not be a power of two (the hash is scaled to it).  Resident wavefronts are
        // what the plans/s of such grids follow (config 3, 4096^2: 2 270 wavefronts of 76 MB 7.5 k plans/s, 3 000 9.0 k;
        // the largest of its 100 000 queries inserts 2.42 M nodes, a shrunk table takes 2.75 M).  FXJPS_TABLE_SHRINK=0:
        // measurement aid.
        if (c.direct_ly == 0 && !getenv("FXJPS_TABLE_LOG2") && !(getenv("FXJPS_TABLE_SHRINK") && atoi(getenv("FXJPS_TABLE_SHRINK")) == 0)) {
            const uint64_t want = std::max<uint64_t>(wpb_up(want_waves), (uint64_t)fx::WPB);
            const double per_entry = (double)sizeof(TEnt) + (1.0 + 1.0 / 8) * sizeof(FarEnt) / 8.0;
            const ScratchCfg& have = d.cfg[0];
            if (have.nbuckets != 0u && have.direct_ly == 0u && have.nwaves >= want) {
                // (the pool in place serves this batch: its size stays -- sizes that follow the free memory of the
                // moment would have the pool allocated and wiped again and again)
                entries = (uint64_t)have.nbuckets * (uint64_t)fx::BUCKET;
                c.usable = have.usable;
            } else if ((double)want * (double)entries * per_entry > (double)budget) {
                uint64_t fit = (uint64_t)((double)budget / ((double)want * per_entry));
                fit = std::max<uint64_t>(fit, entries / 4 * 3) & ~(uint64_t)(8 * fx::BUCKET - 1);
                if (fit < entries) {
                    entries = fit;
                    c.usable = (uint32_t)(entries / 8 * 7);
                }
            }
        }
        c.nbuckets = (uint32_t)(entries / (uint64_t)fx::BUCKET);
        c.far_cap = std::max<uint32_t>(2048u, (uint32_t)(entries / 8));
        if (const char* e = getenv("FXJPS_FAR_CAP")) c.far_cap = (uint32_t)std::min(std::max(atoi(e), 64), 1 << 24);
        c.nwaves = want_waves;
    } else {
        const uint32_t l2e = std::max(ceil_log2(cells * 2 + 64), 12u);
        c.nbuckets = (uint32_t)(((uint64_t)1 << l2e) / (uint64_t)fx::BUCKET);
        c.usable = (uint32_t)(((uint64_t)1 << l2e) / 4 * 3);
        c.far_cap = (uint32_t)std::min<uint64_t>(cells * 2 + 1024, 0x7FFFFFFFull);
        c.nwaves = want_waves;
    }
    c.far_cap = (c.far_cap + 7u) & ~7u;  // the u16 cell-info array behind the entries stays 16-byte granular
    const size_t per_wave = ((size_t)fx::BUCKET * c.nbuckets) * sizeof(TEnt) + (size_t)(c.far_cap + c.far_cap / 8) * sizeof(FarEnt);
    uint32_t maxw = (uint32_t)std::min<size_t>(budget / per_wave, 1u << 20);
    maxw = (uint32_t)wpb_down(maxw);
    if (maxw < (uint32_t)fx::WPB) return fail(h, FXJPS_E_NOMEM, "grid %dx%d needs %zu bytes of scratch per wavefront", d.W, d.H, per_wave);
    if ((uint32_t)wpb_up(c.nwaves) > maxw) d.waves_short = true;
    c.nwaves = std::max((uint32_t)fx::WPB, std::min((uint32_t)wpb_up(c.nwaves), maxw));
    ScratchCfg& cur = d.cfg[pool];
    const bool same = cur.nbuckets == c.nbuckets && cur.usable == c.usable && cur.direct_ly == c.direct_ly && cur.far_cap == c.far_cap && cur.nwaves >= c.nwaves;
    if (same && d.pool_clean[pool]) return FXJPS_OK;
    if (!same) {
        // the old buffers die inside ensure(): forget the old configuration first, so that a failed allocation can
        // never leave a stale cfg pointing at freed (or never wiped) memory
        cur = ScratchCfg();
        d.pool_clean[pool] = false;
        for (;;) {  // on out-of-memory run with fewer resident wavefronts instead of failing the batch
            hipError_t e = d.tables[pool].ensure((size_t)c.nwaves * ((size_t)fx::BUCKET * c.nbuckets));
            if (e == hipSuccess) e = d.far[pool].ensure((size_t)c.nwaves * (c.far_cap + c.far_cap / 8));
            if (e == hipSuccess) e = d.wave_gen[pool].ensure((size_t)c.nwaves);
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            d.tables[pool].release();
            d.far[pool].release();
            if (e != hipErrorOutOfMemory || c.nwaves <= (uint32_t)fx::WPB)
                return fail(h, e == hipErrorOutOfMemory ? FXJPS_E_NOMEM : FXJPS_E_HIP, "scratch pool %d: %s", pool, hipGetErrorString(e));
            c.nwaves = std::max((uint32_t)fx::WPB, (uint32_t)wpb_down(c.nwaves / 2u));
            d.waves_short = true;
            DBG("pool %d: out of memory, retrying with %u wavefronts", pool, c.nwaves);
        }
        cur = c;
    }
    // tables must start all-empty (key 0xFFFFFFFF); wavefronts leave them clean after each query
    HIPCHK(h, hipMemsetAsync(d.tables[pool].p, 0xFF, (size_t)cur.nwaves * ((size_t)fx::BUCKET * cur.nbuckets) * sizeof(TEnt),
                             d.stream));
    HIPCHK(h, hipMemsetAsync(d.wave_gen[pool].p, 0, (size_t)cur.nwaves * sizeof(uint32_t), d.stream));
    d.pool_clean[pool] = true;
    return FXJPS_OK;
}

void fill_search_args(const DevCtx& d, int pool, SearchArgs& A, const uint32_t* d_order, uint32_t nrun, int max_len, bool track);
int launch_search_args(fxjps* h, DevCtx& d, int pool, SearchArgs& A, const ScratchCfg& c, const uint32_t* d_order, uint32_t nrun, int hchoice, bool track);

int launch_search(fxjps* h, DevCtx& d, int pool, const uint32_t* d_order, uint32_t nrun, int hchoice, int max_len, bool track = false) {
    const ScratchCfg& c = d.cfg[pool];
    SearchArgs A;
    fill_search_args(d, pool, A, d_order, nrun, max_len, track);
    return launch_search_args(h, d, pool, A, c, d_order, nrun, hchoice, track);
}

void fill_search_args(const DevCtx& d, int pool, SearchArgs& A, const uint32_t* d_order, uint32_t nrun, int max_len, bool track) {
    const ScratchCfg& c = d.cfg[pool];
    A.G = grid_of(d);
    A.starts = d.d_starts.p;
    A.goals = d.d_goals.p;
    A.order = d_order;
    A.nrun = nrun;
    A.max_len = max_len;
    A.out_path = d.d_path.p;
    A.out_len = d.d_len.p;
    A.out_cost = d.d_cost.p;
    A.out_counters = d.d_counters.p;
    A.qstat = d.d_qstat.p;  // nullptr unless FXJPS_QSTAT is set
    A.qread = track ? d.d_qread.p : nullptr;
    A.tables = d.tables[pool].p;
    A.far = d.far[pool].p;
    A.nbuckets = c.nbuckets;
    A.tab_usable = c.usable;
    A.direct_ly = c.direct_ly;
    // The far band of the open list in f bands that a refill takes whole (no scan, no compaction): pays where open lists
    // hold thousands of entries (4096^2: + 10 %), costs where they hold hundreds (1024^2: - 10 %, an LDS atomic and a
    // scattered store per push instead of an append at a rank).  FXJPS_BANDED=0/1: measurement / test aid.
    A.banded = (uint64_t)d.W * (uint64_t)d.H >= (1ull << 22) ? 1u : 0u;
    if (const char* e = getenv("FXJPS_BANDED")) A.banded = atoi(e) != 0 ? 1u : 0u;
    if (pool != 0) A.banded = 0u;  // the large pool is the last resort: its far tier has no regions that could fill up
    A.far_cap = c.far_cap;
    A.near_max = 512;  // near band of the far tier: re-banded beyond this many entries (FXJPS_NEAR_MAX: test / measurement aid;
                       // measured on c2 / c4 shard / c3: 256 .. 512 with 10 .. 16 refill widths per band is the plateau)
    if (const char* e = getenv("FXJPS_NEAR_MAX")) A.near_max = (uint32_t)std::max(1, atoi(e));
    A.next = d.d_next.p;
    A.wave_gen = d.wave_gen[pool].p;
    A.max_pops = 64ull * (unsigned long long)d.W * d.H + 4096ull;
}

int launch_search_args(fxjps* h, DevCtx& d, int pool, SearchArgs& A, const ScratchCfg& c, const uint32_t* d_order, uint32_t nrun, int hchoice, bool track) {
    uint32_t waves = std::min<uint32_t>(c.nwaves, (nrun + 0u));
    waves = std::max<uint32_t>((uint32_t)fx::WPB, (uint32_t)wpb_up(waves));
    waves = std::min<uint32_t>(waves, c.nwaves);
    // Wavefronts per CU.  A query is a chain of dependent pops, and a wavefront that shares its CU with fifteen others runs
    // that chain more slowly than one that has the CU to itself (config 2, the long diagonal queries: 0.69 us per pop among
    // the batch, 0.62 alone on a CU, 0.59 alone on the chip).  A batch lasts as long as its slowest query; when the batch is
    // small enough for that to show (up to 32 768 queries: beyond, the rest of the batch outlasts every single query), the
    // head of the longest-first order (`nsolo` queries) goes first, in a launch of its own on a second stream: ONE live
    // wavefront per block and an LDS size that lets no second block onto the CU.  The batch's launch is queued when every
    // block of that one has reported from its CU (a counter in pinned memory the host waits on) and fills the rest of the chip.
    // Measured on config 2: 102 k -> 114 k plans/s with 8 .. 48 such queries, two live wavefronts per CU 110 k, four 107 k;
    // on a 125 000-query batch 24 of them cost 1 - 2 %.  FXJPS_SOLO / FXJPS_SOLO_LIVE: measurement and test aids.
    // FXJPS_SPREAD=n (off by default: measured on config 5, 1000 queries, no gain) spreads a batch of at most n queries
    // per CU the same way.
    uint32_t nsolo = (nrun <= 32768u && d.share * h->mem_div == 1) ? 16u : 0u, live_solo = 1, live_main = 0;
    if (const char* e = getenv("FXJPS_SOLO")) nsolo = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("FXJPS_SOLO_LIVE")) live_solo = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("FXJPS_SPREAD")) live_main = (uint32_t)std::max(0, atoi(e));
    if (live_solo != 1u && live_solo != 2u && live_solo != 4u) live_solo = 4u;
    if (live_main != 0u && live_main != 1u && live_main != 2u && live_main != 4u) live_main = 4u;
    if (pool != 0 || track || d_order == nullptr || d.solo_started == nullptr || d.solo_timeouts_run >= 3 || nrun < 4096u || nrun < 64u * nsolo ||
        waves <= 2u * nsolo + (uint32_t)fx::WPB)
        nsolo = 0;
    nsolo = std::min<uint32_t>(nsolo, 512u) & ~(live_solo - 1u);
    if (nsolo != 0u) waves = std::min<uint32_t>(waves, (uint32_t)wpb_down(c.nwaves - nsolo));

    if (pool != 0 || nsolo != 0u || (uint64_t)nrun > (uint64_t)d.n_cu * live_main || d.share * h->mem_div > 1) live_main = 0u;
    // One query per BLOCK (k_search_coop: a searching wavefront and a stager that keeps the LDS tier of its open list in
    // shape, two SIMDs of a CU): for what a handful of long queries decide -- the head launch above, and batches small
    // enough for every query to get a block at once (the frames of config 5, single calls).  Tables indexed by the cell
    // only (grids of up to 2^20 slots), no read-set recording.  FXJPS_COOP=0 / FXJPS_COOP_MAX=n: measurement and test aids.
    uint32_t coop_max = 1024u;
    if (const char* e = getenv("FXJPS_COOP_MAX")) coop_max = (uint32_t)std::max(0, atoi(e));
    // (measured in round 4 and left OFF: the stager answers within ~230 cycles of being asked and takes two refills in three
    // off the searching wavefront -- but a refill turned out to cost that wavefront ~2 300 cycles, not the 6 000 the
    // instrumented build had shown, and taking a block plus re-inserting the late list costs about as much: query 9206
    // alone 69.0 -> 70.7 ms, config 2 130.7 k -> 123 k plans/s with the head launch on such blocks, a config-5 frame
    // 64.5 -> 64.0 ms.  FXJPS_COOP=1 switches it on; tests/test_gpu_fullsize.py::test_cooperative_blocks keeps it exact.)
    const bool coop_ok = pool == 0 && !track && c.direct_ly > 0 && getenv("FXJPS_COOP") && atoi(getenv("FXJPS_COOP")) != 0;
    const bool coop_all = coop_ok && nsolo == 0u && live_main == 0u && nrun <= coop_max && nrun <= c.nwaves;
    if (coop_all) waves = std::min<uint32_t>(c.nwaves, nrun);  // blocks, one scratch slot each
    // A library built with -DFXJPS_XCC=1 and FXJPS_HEAD_XCC=1 in the environment (measurement: see DESIGN.md section 3.1c,
    // `make libfxjps_xcc.so`): the head launch on XCD 0 alone -- an L2 of its own --
    // and the batch's launch on the other seven.  Needs the round-robin rule "block b runs on XCD b % 8" (probed once).
    bool xcc_split = FXJPS_XCC != 0 && nsolo != 0u && live_solo == 1u && !coop_ok && d.n_cu == 256 && getenv("FXJPS_HEAD_XCC") && atoi(getenv("FXJPS_HEAD_XCC")) != 0;
    if (xcc_split && d.xcc_rr < 0) {
        d.xcc_rr = 0;
        uint32_t* dp = nullptr;
        uint32_t hp[64];
        if (hipMalloc((void**)&dp, sizeof(hp)) == hipSuccess) {
            hipLaunchKernelGGL(fx::k_xcc_probe, dim3(64), dim3(64), 0, d.stream, dp);
            if (hipMemcpyAsync(hp, dp, sizeof(hp), hipMemcpyDeviceToHost, d.stream) == hipSuccess && hipStreamSynchronize(d.stream) == hipSuccess) {
                d.xcc_rr = 1;
                for (int b = 0; b < 64; b++)
                    if (hp[b] != (uint32_t)(b % 8)) d.xcc_rr = 0;
            }
            (void)hipFree(dp);
        }
        (void)hipGetLastError();
    }
    if (xcc_split && (d.xcc_rr != 1 || nsolo > 32u)) xcc_split = false;
    if (xcc_split) waves = std::min<uint32_t>(waves, (uint32_t)(d.n_cu / 8 * 7) * 4u * (uint32_t)fx::OCC);  // what seven XCDs hold at once
    if (pool == 0) d.waves_used = waves;
    HIPCHK(h, hipMemsetAsync(d.d_next.p, 0, 4 * sizeof(unsigned int), d.stream));
    const dim3 block(fx::WAVE * fx::WPB);
    DBG("launch k_search pool=%d waves=%u nrun=%u buckets=%u far_cap=%u solo=%u x %u spread=%u coop=%d", pool, waves, nrun, c.nbuckets, c.far_cap, nsolo,
        live_solo, live_main, coop_all ? 2 : (coop_ok ? 1 : 0));
    HIPCHK(h, hipEventRecord(d.ev0, d.stream));
    // instantiations: heuristic x read-set recording (fxjps_replan_frame) x table indexed by the cell
    {
        using KFn = void (*)(SearchArgs);
        static const KFn kfn[2][2][2] = {{{fx::k_search<1, false, false>, fx::k_search<1, false, true>},
                                          {fx::k_search<1, true, false>, fx::k_search<1, true, true>}},
                                         {{fx::k_search<2, false, false>, fx::k_search<2, false, true>},
                                          {fx::k_search<2, true, false>, fx::k_search<2, true, true>}}};
        const KFn fn = kfn[hchoice == 1 ? 0 : 1][track ? 1 : 0][c.direct_ly > 0 ? 1 : 0];
        static const size_t pad = 36u << 10;  // 66 .. 74 KB of the block's own + this: no second block fits the CU's 160 KB
        if (nsolo != 0u || live_main != 0u) {
            const int ki = (hchoice == 1 ? 0 : 4) + (track ? 2 : 0) + (c.direct_ly > 0 ? 1 : 0);
            if (!d.lds_attr_done[ki]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad) == hipSuccess) {
                    d.lds_attr_done[ki] = true;
                } else {  // no such launches on this device then: the batch runs as one launch (speed, not correctness)
                    (void)hipGetLastError();
                    nsolo = 0u;
                    live_main = 0u;
                }
            }
        }
        const KFn cfn = hchoice == 1 ? fx::k_search_coop<1> : fx::k_search_coop<2>;
        static const size_t cpad = 100u << 10;  // the head launch's blocks: 21 KB of their own + this: no second block, and none of the batch's (66 KB), fits the CU
        bool coop_head = coop_ok && nsolo != 0u && live_solo == 1u;
        if (coop_head && !d.lds_attr_coop[hchoice == 1 ? 0 : 1]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(cfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cpad) == hipSuccess) {
                d.lds_attr_coop[hchoice == 1 ? 0 : 1] = true;
            } else {
                (void)hipGetLastError();
                coop_head = false;
            }
        }
        if (nsolo != 0u) {
            SearchArgs B = A;
            B.nrun = nsolo;
            B.solo = live_solo;
            B.wave_base = waves;
            B.next = d.d_next.p + 1;
            A.q0 = nsolo;
            A.nrun = nrun - nsolo;
            B.started = d.solo_started;
            if (xcc_split) {  // eight times the blocks: every eighth one finds itself on XCD 0 and works
                B.xcc_only = 1u;
                B.slot_ctr = d.d_next.p + 3;
                B.max_blocks = nsolo;
                A.xcc_only = 0xFEu;
                A.slot_ctr = d.d_next.p + 2;
                A.max_blocks = waves / (uint32_t)fx::WPB;
            }
            HIPCHK(h, hipEventRecord(d.ev_solo0, d.stream));
            HIPCHK(h, hipStreamWaitEvent(d.stream_solo, d.ev_solo0, 0));
            HIPCHK(h, hipEventRecord(d.ev_hd0, d.stream_solo));
            if (coop_head)
                hipLaunchKernelGGL(cfn, dim3(nsolo), dim3(fx::WAVE * 2), cpad, d.stream_solo, B);
            else
            hipLaunchKernelGGL(fn, dim3((nsolo / live_solo) * (xcc_split ? 8u : 1u)), block, pad, d.stream_solo, B);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipEventRecord(d.ev_hd1, d.stream_solo));
            HIPCHK(h, hipEventRecord(d.ev_solo1, d.stream_solo));
            // The batch's launch must not take the CUs first: it is queued when every block of this one has reported from
            // its CU (a counter in pinned host memory; tens of microseconds).  The host waits, not the stream: a stream
            // wait on a value only another queue's kernel can write deadlocks under tools that run one kernel at a time
            // (rocprofv3 --pmc).  After 5 ms the batch goes ahead regardless -- placement is speed, never correctness.
            d.solo_target += nsolo / live_solo;
            const auto t0 = std::chrono::steady_clock::now();
            bool late = false;
            while ((int32_t)(__atomic_load_n(d.solo_started, __ATOMIC_ACQUIRE) - d.solo_target) < 0) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
                    // (the counter did not arrive in time -- no PCIe atomics, a tool that serialises kernels: after three
                    // such waits in a row this device runs its batches as one launch; fxjps_timing_t::solo_timeouts counts them)
                    late = true;
                    break;
                }
            }
            d.solo_timeouts += late ? 1 : 0;
            d.solo_timeouts_run = late ? d.solo_timeouts_run + 1 : 0;
        }
        if (nsolo != 0u) HIPCHK(h, hipEventRecord(d.ev_bt0, d.stream));
        if (coop_all) {
            hipLaunchKernelGGL(cfn, dim3(waves), dim3(fx::WAVE * 2), 0, d.stream, A);
        } else if (live_main != 0u) {
            A.solo = live_main;
            hipLaunchKernelGGL(fn, dim3(waves / live_main), block, pad, d.stream, A);
        } else {
            // (split over the XCDs: as many blocks as the whole chip holds, an eighth of them leave at once)
            hipLaunchKernelGGL(fn, dim3(xcc_split ? (unsigned)(d.n_cu * 2) : waves / fx::WPB), block, 0, d.stream, A);
        }
        HIPCHK(h, hipGetLastError());
        if (nsolo != 0u) {
            HIPCHK(h, hipEventRecord(d.ev_bt1, d.stream));
            HIPCHK(h, hipStreamWaitEvent(d.stream, d.ev_solo1, 0));
        }
        if (pool == 0) d.had_solo = nsolo != 0u;
    }
    HIPCHK(h, hipEventRecord(d.ev1, d.stream));
    d.launches += nsolo != 0u ? 2 : 1;  // (the two launches overlap: the events span both)
    return FXJPS_OK;
}

// Runs the shard [q0, q0+nq) of the batch on device d up to the point where
// len/cost/offsets are in pinned host memory and the CSR cells are packed on
// the device.
int run_shard(fxjps* h, DevCtx& d, const int32_t* starts, const int32_t* goals, int hchoice, int max_len) {
    HIPCHK(h, hipSetDevice(d.dev));
    const int64_t nq = d.nq;
    d.kernel_ms = 0;
    d.launches = 0;
    d.retried = 0;
    d.nrun = 0;
    d.waves_used = 0;
    d.waves_short = false;
    d.had_solo = false;
    if (nq == 0) return FXJPS_OK;
    HIPCHK(h, d.d_starts.ensure((size_t)nq * 2));
    HIPCHK(h, d.d_goals.ensure((size_t)nq * 2));
    HIPCHK(h, d.d_len.ensure((size_t)nq));
    HIPCHK(h, d.d_cost.ensure((size_t)nq));
    d.cells_bound = (size_t)nq * (size_t)max_len * 2 * sizeof(int32_t);
    if (d.d_path.ensure((size_t)nq * max_len) != hipSuccess) {  // the scratch pool of an earlier, smaller batch may be in the way
        (void)hipGetLastError();
        release_pool0(d);
        HIPCHK(h, d.d_path.ensure((size_t)nq * max_len));
    }
    HIPCHK(h, d.d_offsets.ensure((size_t)nq + 1));
    HIPCHK(h, d.d_next.ensure(4));
    HIPCHK(h, d.d_counters.ensure(64));
    HIPCHK(h, d.h_len.ensure((size_t)nq));
    HIPCHK(h, d.h_cost.ensure((size_t)nq));
    HIPCHK(h, d.h_offsets.ensure((size_t)nq + 1));
    HIPCHK(h, d.h_counters.ensure(64));
    HIPCHK(h, hipMemcpyAsync(d.d_starts.p, starts + 2 * d.q0, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
    HIPCHK(h, hipMemcpyAsync(d.d_goals.p, goals + 2 * d.q0, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
    HIPCHK(h, hipMemsetAsync(d.d_counters.p, 0, 64 * sizeof(unsigned long long), d.stream));
    if (getenv("FXJPS_QSTAT")) {  // diagnostics: per-query start / end time, pops, wavefront (tools/qstat.py)
        HIPCHK(h, d.d_qstat.ensure((size_t)nq * 4));
        HIPCHK(h, hipMemsetAsync(d.d_qstat.p, 0, (size_t)nq * 4 * sizeof(unsigned long long), d.stream));
    }
    uint32_t full = (uint32_t)d.n_cu * 4u * (uint32_t)fx::OCC / (uint32_t)(d.share * h->mem_div);  // every wavefront the chip can hold at once (this handle's share of them)
    full = std::max<uint32_t>((uint32_t)wpb_down(full), (uint32_t)fx::WPB);
    if (const char* e = getenv("FXJPS_WAVES")) full = (uint32_t)wpb_down((uint64_t)std::max(fx::WPB, atoi(e)));  // measurement aid
    // Longest-processing-time-first: the time a query takes grows with the start-goal distance, so far-apart queries are
    // handed out first and the short ones fill the tail.  The key is max(dx, dy) + min(dx, dy) / 2: on the config-2
    // workload it correlates 0.95 with the time a query takes and 0.95 with its expansions (Chebyshev distance: 0.91 /
    // 0.94; Manhattan: 0.94 / 0.92) -- the queries that end last are the long DIAGONAL ones, whose open lists are full of
    // equal keys (fewer nodes committed per iteration).  Counting sort, descending.
    {
        const int32_t* S = starts + 2 * d.q0;
        const int32_t* G = goals + 2 * d.q0;
        constexpr uint32_t KMAX = 3u * 8192u - 1u;
        std::vector<uint32_t> head(KMAX + 3u, 0);
        d.h_order.resize((size_t)nq);
        auto key = [&](int64_t i) -> uint32_t {
            const int64_t dx = std::llabs((int64_t)S[2 * i] - G[2 * i]), dy = std::llabs((int64_t)S[2 * i + 1] - G[2 * i + 1]);
            return (uint32_t)std::min<int64_t>(2 * std::max(dx, dy) + std::min(dx, dy), (int64_t)KMAX);
        };
        const bool subset = d.mode == 2;  // streaming replan: only the queries whose read set was touched
        auto in = [&](int64_t i) -> bool { return !subset || d.h_sel[(size_t)i] != 0; };
        for (int64_t i = 0; i < nq; i++)
            if (in(i)) head[KMAX - key(i) + 1]++;
        for (uint32_t k = 1; k < KMAX + 3u; k++) head[k] += head[k - 1];
        d.nrun = head[KMAX + 2u];
        for (int64_t i = 0; i < nq; i++)
            if (in(i)) d.h_order[head[KMAX - key(i)]++] = (uint32_t)i;
        // Measurement aid (round 6, DESIGN.md section 4: no gain, off): within classes of 128 key values the queries in Morton
        // order of the middle of start and goal, so that wavefronts that start together read neighbouring lines of the maps.
        // The order is not an output: results are the same bytes either way.
        static const bool morton = getenv("FXJPS_ORDER_MORTON") && atoi(getenv("FXJPS_ORDER_MORTON")) != 0;
        if (morton && d.nrun > 1) {
            auto spread = [](uint32_t v) -> uint64_t {  // bits of v to the even positions
                uint64_t x = v & 0xFFFFu;
                x = (x | (x << 8)) & 0x00FF00FFull;
                x = (x | (x << 4)) & 0x0F0F0F0Full;
                x = (x | (x << 2)) & 0x33333333ull;
                x = (x | (x << 1)) & 0x55555555ull;
                return x;
            };
            std::vector<std::pair<uint64_t, uint32_t>> ks((size_t)d.nrun);
            for (int64_t j = 0; j < d.nrun; j++) {
                const int64_t i = d.h_order[(size_t)j];
                const uint32_t mx = (uint32_t)((S[2 * i] + G[2 * i]) / 2), my = (uint32_t)((S[2 * i + 1] + G[2 * i + 1]) / 2);
                ks[(size_t)j] = {((uint64_t)((KMAX - key(i)) >> 7) << 32) | (spread(mx >> 3) << 1) | spread(my >> 3), (uint32_t)i};
            }
            std::sort(ks.begin(), ks.end());
            for (int64_t j = 0; j < d.nrun; j++) d.h_order[(size_t)j] = ks[(size_t)j].second;
        }
        HIPCHK(h, d.d_order.ensure((size_t)nq));
        if (d.nrun > 0)
            HIPCHK(h, hipMemcpyAsync(d.d_order.p, d.h_order.data(), (size_t)d.nrun * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    }
    if (d.mode != 0) {  // read-set bitmaps: 128 x u64 per query, zero for queries that never searched
        const size_t had = d.d_qread.cap;
        HIPCHK(h, d.d_qread.ensure((size_t)nq * 128));
        if (d.d_qread.cap != had) HIPCHK(h, hipMemsetAsync(d.d_qread.p, 0, d.d_qread.cap * sizeof(unsigned long long), d.stream));
    }
    if (d.nrun == 0) return FXJPS_OK;  // every result of the previous frame is still valid
    DBG("run_shard nq=%lld: inputs queued", (long long)nq);
    int rc = ensure_pool(h, d, 0, (uint32_t)std::min<int64_t>(full, (d.nrun + 3) & ~3ll));
    if (rc) return rc;
    DBG("pool ready");
    rc = launch_search(h, d, 0, d.d_order.p, (uint32_t)d.nrun, hchoice, max_len, d.mode != 0);
    if (rc) return rc;
    return FXJPS_OK;
}

// Second half: wait for the search, retry overflowed queries with the large
// pool, pack to CSR, copy len/cost/offsets back.
int finish_shard(fxjps* h, DevCtx& d, int hchoice, int max_len) {
    HIPCHK(h, hipSetDevice(d.dev));
    const int64_t nq = d.nq;
    if (nq == 0) return FXJPS_OK;
    // The result lengths are copied AFTER the host has seen the search finish, not queued behind it: a copy that waits
    // in the copy engine's queue for a kernel holds up every copy queued after it -- the input copies of another handle's
    // next batch on this device among them, and with them that batch's launch (two handles planning config-2 batches in
    // turn ran strictly one after the other: tools/copy_timeline.py).
    DBG("waiting for the search kernel");
    HIPCHK(h, hipStreamSynchronize(d.stream));
    DBG("search kernel done");
    HIPCHK(h, hipMemcpyAsync(d.h_len.p, d.d_len.p, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipStreamSynchronize(d.stream));
    float ms = 0;
    d.head_ms = d.batch_ms = 0;
    if (d.launches > 0) {
        HIPCHK(h, hipEventElapsedTime(&ms, d.ev0, d.ev1));
        d.kernel_ms += ms;
        if (d.had_solo) {
            float a = 0, b = 0;
            HIPCHK(h, hipEventElapsedTime(&a, d.ev_hd0, d.ev_hd1));
            HIPCHK(h, hipEventElapsedTime(&b, d.ev_bt0, d.ev_bt1));
            d.head_ms = a;
            d.batch_ms = b;
        }
    }
    std::vector<uint32_t> redo;
    for (int64_t i = 0; i < nq; i++)
        if (d.h_len.p[i] <= fx::QI_TABLE_FULL) redo.push_back((uint32_t)i);
    if (!redo.empty()) {
        d.retried = (int64_t)redo.size();
        int rc = ensure_pool(h, d, 1, (uint32_t)std::min<size_t>(256, (redo.size() + 3) & ~(size_t)3));
        if (rc) return rc;
        HIPCHK(h, d.d_redo.ensure(redo.size()));
        HIPCHK(h, hipMemcpyAsync(d.d_redo.p, redo.data(), redo.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
        HIPCHK(h, hipStreamSynchronize(d.stream));  // `redo` is pageable host memory
        rc = launch_search(h, d, 1, d.d_redo.p, (uint32_t)redo.size(), hchoice, max_len, d.mode != 0);
        if (rc) return rc;
        HIPCHK(h, hipStreamSynchronize(d.stream));  // (as above: no copy waits in the engine's queue for a search)
        HIPCHK(h, hipMemcpyAsync(d.h_len.p, d.d_len.p, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
        HIPCHK(h, hipStreamSynchronize(d.stream));
        HIPCHK(h, hipEventElapsedTime(&ms, d.ev0, d.ev1));
        d.kernel_ms += ms;
    }
    DBG("kernel %.3f ms, %zu to redo; scan", d.kernel_ms, redo.size());
    hipLaunchKernelGGL(fx::k_scan_len, dim3(1), dim3(1024), 0, d.stream, d.d_len.p, (long long)nq, d.d_offsets.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(d.h_offsets.p, d.d_offsets.p, ((size_t)nq + 1) * sizeof(long long), hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipMemcpyAsync(d.h_cost.p, d.d_cost.p, (size_t)nq * sizeof(double), hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipMemcpyAsync(d.h_counters.p, d.d_counters.p, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, d.stream));
    if (d.mode != 0 && d.launches > 0) {  // the read sets of what ran (the rest of the buffer is unchanged)
        d.h_qread.resize((size_t)nq * 128);
        HIPCHK(h, hipMemcpyAsync(d.h_qread.data(), d.d_qread.p, (size_t)nq * 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost, d.stream));
    }
    HIPCHK(h, hipStreamSynchronize(d.stream));
    const long long total = d.h_offsets.p[nq];
    DBG("scan done, %lld cells", total);
    if (total > 0) {
        if (d.d_cells.ensure((size_t)total * 2) != hipSuccess) {  // (the search is over: its scratch can go)
            (void)hipGetLastError();
            release_pool0(d);
            HIPCHK(h, d.d_cells.ensure((size_t)total * 2));
        }
        HIPCHK(h, d.h_cells.ensure((size_t)total * 2));
        hipLaunchKernelGGL(fx::k_gather_paths, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, d.stream, d.d_path.p,
                           d.d_len.p, d.d_offsets.p, (long long)nq, max_len, d.d_cells.p, total);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(d.h_cells.p, d.d_cells.p, (size_t)total * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
        HIPCHK(h, hipStreamSynchronize(d.stream));
    }
    DBG("gather done");
    // internal codes must not leak
    for (int64_t i = 0; i < nq; i++)
        if (d.h_len.p[i] <= fx::QI_TABLE_FULL) d.h_len.p[i] = FXJPS_Q_CAPACITY;
    return FXJPS_OK;
}

// One query per call -- the node's real use of the drop-in (global_planner_st.py:285: jps1.method once per tick, on maps
// of ~150 x 110 cells).  The batch machinery around the search (uploads of the query arrays and of the longest-first
// order, the work counter, the length scan and the CSR gather with their three host waits) costs more than such a search:
// here the start and the goal travel in the kernel arguments, ONE wavefront runs the query, and the kernel writes the
// path, its length and its cost straight into the handle's pinned host buffers -- one launch, one host wait.  Results
// land where plan_core's other path leaves them (h_len / h_cost / h_offsets / h_cells), so the callers above see no
// difference.  Returns 1 when the query has to go through the batch path after all (it outgrew the regular scratch).
int plan_single(fxjps* h, DevCtx& d, const int32_t* starts, const int32_t* goals, int hchoice, int max_len) {
    HIPCHK(h, hipSetDevice(d.dev));
    d.q0 = 0;
    d.nq = 1;
    d.mode = 0;
    d.kernel_ms = 0;
    d.launches = 0;
    d.retried = 0;
    d.nrun = 1;
    d.waves_used = 1;
    d.waves_short = false;
    d.had_solo = false;
    d.head_ms = d.batch_ms = 0;
    HIPCHK(h, d.h_len.ensure(1));
    HIPCHK(h, d.h_cost.ensure(1));
    HIPCHK(h, d.h_offsets.ensure(2));
    HIPCHK(h, d.h_counters.ensure(64));
    HIPCHK(h, d.d_counters.ensure(64));
    HIPCHK(h, d.h_path1.ensure((size_t)max_len));
    HIPCHK(h, d.h_cells.ensure((size_t)max_len * 2));
    d.cells_bound = 0;
    int rc = ensure_pool(h, d, 0, (uint32_t)fx::WPB);
    if (rc) return rc;
    const ScratchCfg& c = d.cfg[0];
    SearchArgs A;
    fill_search_args(d, 0, A, nullptr, 1u, max_len, false);
    A.starts = nullptr;
    A.goals = nullptr;
    A.next = nullptr;
    A.qstat = nullptr;
    A.single = 1u;
    A.solo = 1u;
    A.imm[0] = starts[0];
    A.imm[1] = starts[1];
    A.imm[2] = goals[0];
    A.imm[3] = goals[1];
    void *dp_path = nullptr, *dp_len = nullptr, *dp_cost = nullptr;
    HIPCHK(h, hipHostGetDevicePointer(&dp_path, d.h_path1.p, 0));
    HIPCHK(h, hipHostGetDevicePointer(&dp_len, d.h_len.p, 0));
    HIPCHK(h, hipHostGetDevicePointer(&dp_cost, d.h_cost.p, 0));
    A.out_path = (uint32_t*)dp_path;
    A.out_len = (int32_t*)dp_len;
    A.out_cost = (double*)dp_cost;
    d.h_len.p[0] = fx::QI_WATCHDOG;  // (overwritten by the kernel)
    // (round 6: the kernel zeroes its 64 counters itself and leaves them in the pinned host buffer when it is done -- a memset
    // and a copy back were two more operations for the runtime to queue, ~ 12 us of a call)
    void* dp_cnt = nullptr;
    HIPCHK(h, hipHostGetDevicePointer(&dp_cnt, d.h_counters.p, 0));
    A.host_counters = (unsigned long long*)dp_cnt;
    {
        using KFn = void (*)(SearchArgs);
        static const KFn kfn[2][2] = {{fx::k_search<1, false, false>, fx::k_search<1, false, true>},
                                      {fx::k_search<2, false, false>, fx::k_search<2, false, true>}};
        hipLaunchKernelGGL(kfn[hchoice == 1 ? 0 : 1][c.direct_ly > 0 ? 1 : 0], dim3(1), dim3(fx::WAVE * fx::WPB), 0, d.stream, A);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipStreamSynchronize(d.stream));
    // (the kernel's time by the device's constant-rate clock, read by the one wavefront at its first and last instruction: two
    // event records and hipEventElapsedTime were three more runtime calls on a 250 us call)
    d.kernel_ms = (double)(d.h_counters.p[63] - d.h_counters.p[62]) / (double)d.wall_khz;
    DBG("single call: kernel %.1f us, %llu pops, shader clock %.0f MHz", d.kernel_ms * 1e3, (unsigned long long)d.h_counters.p[0],
        d.kernel_ms > 0 ? (double)(d.h_counters.p[61] - d.h_counters.p[60]) / (d.kernel_ms * 1e3) : 0.0);
    d.h_counters.p[60] = d.h_counters.p[61] = d.h_counters.p[62] = d.h_counters.p[63] = 0ull;
    d.launches = 1;
    const int32_t n = d.h_len.p[0];
    if (n <= fx::QI_TABLE_FULL) return 1;  // outgrew the regular scratch (or the watchdog): the batch path has the large pool
    d.h_offsets.p[0] = 0;
    d.h_offsets.p[1] = n > 0 ? n : 0;
    for (int32_t i = 0; i < n; i++) {
        const uint32_t v = d.h_path1.p[i];
        d.h_cells.p[2 * i] = (int32_t)(v >> 16);
        d.h_cells.p[2 * i + 1] = (int32_t)(v & 0xFFFFu);
    }
    return FXJPS_OK;
}

int update_cells_async(fxjps* h, const int32_t* xy, const uint8_t* val, int64_t n, bool derive);  // (below, with the streaming entry points)
void drain_all(fxjps* h);

int plan_core(fxjps* h, const int32_t* starts, const int32_t* goals, int64_t nq, int hchoice, int max_len, int mode = 0) {
    if (!h) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "fxjps_plan_batch before fxjps_set_grid");
    if (nq < 0 || (nq > 0 && (!starts || !goals))) return fail(h, FXJPS_E_ARG, "bad query arrays");
    if (hchoice != 1 && hchoice != 2)
        return fail(h, FXJPS_E_ARG, "hchoice must be 1 or 2 (the reference raises TypeError, jps1.py:188)");
    if (max_len < 1 || max_len > (1 << 20)) return fail(h, FXJPS_E_ARG, "max_path_len out of range");
    if (nq > 0x7FFFFFF0ll) return fail(h, FXJPS_E_ARG, "too many queries in one batch");
    // Until this batch is complete (emit_csr) no earlier batch is "the last batch": the resident-path forms of the
    // waypoint entry points refuse with FXJPS_E_ARG instead of reading lengths and offsets of different batches.
    h->last_nq = 0;
    if (h->maps_stale) {  // deferred cell updates: the maps are rebuilt once, in front of the search
        int rc = update_cells_async(h, nullptr, nullptr, 0, true);
        if (rc) return rc;
    }
    const int nd = (int)h->devs.size();
    for (int r = 0; r < nd; r++) {  // contiguous shards: SURVEY 8(e)
        h->devs[r].q0 = nq * r / nd;
        h->devs[r].nq = nq * (r + 1) / nd - h->devs[r].q0;
        h->devs[r].mode = mode;
    }
    h->q_results_valid = false;  // (fxjps_replan_frame sets it again once its frame is complete)
    int rc = FXJPS_OK;
    h->last_on_host = false;
    static const bool single_ok = !(getenv("FXJPS_SINGLE") && atoi(getenv("FXJPS_SINGLE")) == 0) && !getenv("FXJPS_QSTAT");  // (0: test / measurement aid)
    bool done = false;
    if (nq == 1 && nd == 1 && mode == 0 && single_ok) {
        rc = plan_single(h, h->devs[0], starts, goals, hchoice, max_len);
        if (rc == 1) {
            rc = FXJPS_OK;  // (rare: on to the batch path)
        } else {
            done = true;
            h->last_on_host = rc == FXJPS_OK;
        }
    }
    if (nd > 1 && !rc && !done) {
        // One host thread per context: each queues its shard's copies and launches, waits for its own search (the head
        // launch's start counter, then the kernel), packs its paths and brings them back.  The devices never wait for each
        // other's host-side tail -- run one after the other, eight tails of ~ 23 ms each behind ~ 570 ms of parallel search
        // capped config 4 at 75 % strong-scaling efficiency by construction.  (The shards share nothing, jps1.py:183-192.)
        rc = run_side_by_side((size_t)nd, [&](size_t r) {
            int e = run_shard(h, h->devs[r], starts, goals, hchoice, max_len);
            if (!e) e = finish_shard(h, h->devs[r], hchoice, max_len);
            return e;
        });
        done = true;
    }
    for (int r = 0; r < nd && !rc && !done; r++) rc = run_shard(h, h->devs[r], starts, goals, hchoice, max_len);
    for (int r = 0; r < nd && !rc && !done; r++) rc = finish_shard(h, h->devs[r], hchoice, max_len);
    if (rc) {
        drain_all(h);  // before the error leaves the library
        for (auto& d : h->devs) d.nq = 0;  // (no shard holds a result)
        return rc;
    }
    fxjps_timing_t& T = h->timing;
    T.search_kernel_ms = 0;
    T.search_launches = 0;
    T.retried = 0;
    T.pops = 0;
    T.pushes = 0;
    T.far_refills = 0;
    T.slow_pops = 0;
    T.table_wipes = 0;
    T.reused = 0;
    T.table_direct = 0;
    T.waves = 0;
    T.waves_short = 0;
    T.head_launch_ms = 0;
    T.batch_launch_ms = 0;
    T.solo_timeouts = 0;
    for (auto& d : h->devs) {
        T.head_launch_ms = std::max(T.head_launch_ms, d.head_ms);
        T.batch_launch_ms = std::max(T.batch_launch_ms, d.batch_ms);
        T.solo_timeouts += d.solo_timeouts;
        T.waves += d.waves_used;
        if (d.waves_short) T.waves_short = 1;
        T.search_kernel_ms = std::max(T.search_kernel_ms, d.kernel_ms);
        T.search_launches += d.launches;
        T.retried += d.retried;
        if (d.nq > 0) {
            T.pops += (int64_t)d.h_counters.p[0];
            T.pushes += (int64_t)d.h_counters.p[1];
            T.far_refills += (int64_t)d.h_counters.p[2];
            T.slow_pops += (int64_t)d.h_counters.p[3];
            T.table_wipes += (int64_t)d.h_counters.p[7];
            if (d.cfg[0].direct_ly > 0) T.table_direct = 1;
        }
    }
    return FXJPS_OK;
}

typedef int (*nccl_init_all_t)(void**, int, const int*);
typedef int (*nccl_bcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_group_t)(void);
typedef int (*nccl_destroy_t)(void*);

typedef struct {
    char b[128];
} nccl_uid_t;  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128
typedef int (*nccl_get_uid_t)(nccl_uid_t*);
typedef int (*nccl_init_rank_t)(void**, int, nccl_uid_t, int);

void* load_rccl(fxjps* h) {
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) fail(h, FXJPS_E_COMM, "cannot load librccl.so: %s", dlerror());
    return lib;
}

int broadcast_grid(fxjps* h, int W, int H) {
    const int nd = (int)h->devs.size();
    if (h->world > 0) {
        // one process per GPU: this rank's part of the one collective of the path -- ncclBroadcast of the W*H occupancy
        // bytes from rank 0 over xGMI, on this rank's stream (every rank calls fxjps_set_grid_rank with the same W, H)
        if (h->world == 1 && !h->rccl) return FXJPS_OK;
        auto bcast = (nccl_bcast_t)dlsym(h->rccl, "ncclBroadcast");
        if (!bcast) return fail(h, FXJPS_E_COMM, "ncclBroadcast not found");
        DevCtx& d = h->devs[0];
        HIPCHK(h, hipSetDevice(d.dev));
        if (bcast(d.occ.p, d.occ.p, (size_t)W * H, /*ncclUint8*/ 1, 0, h->comms[0], d.stream) != 0)
            return fail(h, FXJPS_E_COMM, "ncclBroadcast failed on rank %d", h->rank);
        return FXJPS_OK;
    }
    // (FXJPS_FORCE_RCCL=1: test aid -- a one-device handle goes through the collective too: a communicator of one rank,
    // an in-place broadcast from itself.  Exercises the loading of librccl, the symbols, their signatures and the
    // communicator's life cycle on a one-GPU box.)
    if (nd == 1 && !(getenv("FXJPS_FORCE_RCCL") && atoi(getenv("FXJPS_FORCE_RCCL")) != 0)) return FXJPS_OK;
    bool distinct = true;
    for (int a = 0; a < nd; a++)
        for (int b = a + 1; b < nd; b++)
            if (h->devs[a].dev == h->devs[b].dev) distinct = false;
    if (!distinct) {
        // several contexts share a device (more shards than GPUs): RCCL wants one rank per device, so the grid is
        // handed over by plain copies -- on-device for a context of the root's device, peer-to-peer otherwise
        DevCtx& d0 = h->devs[0];
        for (int r = 1; r < nd; r++) {
            DevCtx& d = h->devs[r];
            if (d.dev == d0.dev) {
                HIPCHK(h, hipSetDevice(d.dev));
                HIPCHK(h, hipMemcpyAsync(d.occ.p, d0.occ.p, (size_t)W * H, hipMemcpyDeviceToDevice, d.stream));
            } else {
                HIPCHK(h, hipMemcpyPeerAsync(d.occ.p, d.dev, d0.occ.p, d0.dev, (size_t)W * H, d.stream));
            }
        }
        return FXJPS_OK;
    }
    if (!h->rccl) {
        h->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h->rccl) h->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h->rccl) return fail(h, FXJPS_E_COMM, "cannot load librccl.so: %s", dlerror());
        auto init_all = (nccl_init_all_t)dlsym(h->rccl, "ncclCommInitAll");
        if (!init_all) return fail(h, FXJPS_E_COMM, "ncclCommInitAll not found");
        std::vector<int> ids;
        for (auto& d : h->devs) ids.push_back(d.dev);
        h->comms.assign(nd, nullptr);
        if (init_all(h->comms.data(), nd, ids.data()) != 0) return fail(h, FXJPS_E_COMM, "ncclCommInitAll failed");
    }
    auto bcast = (nccl_bcast_t)dlsym(h->rccl, "ncclBroadcast");
    auto gstart = (nccl_group_t)dlsym(h->rccl, "ncclGroupStart");
    auto gend = (nccl_group_t)dlsym(h->rccl, "ncclGroupEnd");
    if (!bcast || !gstart || !gend) return fail(h, FXJPS_E_COMM, "RCCL symbols missing");
    // one broadcast of W*H bytes, root = first device, over xGMI; no other collective exists on this path
    gstart();
    for (int r = 0; r < nd; r++) {
        DevCtx& d = h->devs[r];
        (void)hipSetDevice(d.dev);
        if (bcast(d.occ.p, d.occ.p, (size_t)W * H, /*ncclUint8*/ 1, 0, h->comms[r], d.stream) != 0) {
            gend();
            return fail(h, FXJPS_E_COMM, "ncclBroadcast failed on device %d", d.dev);
        }
    }
    if (gend() != 0) return fail(h, FXJPS_E_COMM, "ncclGroupEnd failed");
    return FXJPS_OK;
}

// A handle that fxjps_create_rank made with world > 1 sets its grid with fxjps_set_grid_rank and nothing else: that call
// is a collective (every rank enters ncclBroadcast), so any other whole-grid setter on such a handle would have this rank
// wait in a broadcast its peers never join.
int refuse_on_rank_handle(fxjps* h, const char* what) {
    if (h->world > 1) return fail(h, FXJPS_E_ARG, "%s on rank %d of %d: a rank handle sets its grid with fxjps_set_grid_rank (collective)", what, h->rank, h->world);
    return FXJPS_OK;
}

int finish_set_grid(fxjps* h, int W, int H, bool wait = true) {
    int rc = broadcast_grid(h, W, H);
    if (rc) return rc;
    for (auto& d : h->devs) {
        rc = derive_maps(h, d);
        if (rc) return rc;
    }
    for (auto& d : h->devs) {
        if (!wait) break;  // (one context, a small grid out of a staging buffer: whatever is queued next runs behind the build)
        HIPCHK(h, hipSetDevice(d.dev));
        HIPCHK(h, hipStreamSynchronize(d.stream));
    }
    h->have_grid = true;
    h->maps_stale = false;  // (derived from the grid that was just set: a deferred update of the old grid is moot)
    return FXJPS_OK;
}

// every stream of the handle idle (error paths: copies and kernels of the other devices may still be in flight on the
// caller's buffers and on device buffers the next call may reallocate)
void drain_all(fxjps* h) {
    const std::string keep = h->err;
    for (auto& d : h->devs) {
        if (hipSetDevice(d.dev) == hipSuccess) {
            if (d.stream_solo) (void)hipStreamSynchronize(d.stream_solo);  // (a head launch may still be running)
            (void)hipStreamSynchronize(d.stream);
        }
        d.pool_clean[0] = d.pool_clean[1] = false;  // a search may have died half-way through a table
        d.owner_ready = d.chg_ready = false;        // ... or an update half-way through its marks
    }
    (void)hipGetLastError();
    h->err = keep;
}

// ---- waypoint selection over a batch: where the paths are.  The caller's CSR (device 0 takes all of them), or the last
// batch's, resident shard by shard on the devices that planned them.
struct WpPart {
    DevCtx* d;
    int64_t q0, n;
    const long long* d_off;
    const int32_t *d_cells, *d_len;
    long long total;
};
int wp_gather_parts(fxjps* h, int64_t nq, const int64_t* offsets, const int32_t* cells_xy, bool nonneg, std::vector<WpPart>& parts,
                    std::vector<int64_t>& kept_at, int64_t& kept_base) {
    kept_base = 0;
    if (cells_xy) {
        for (int64_t q = 0; q < nq; q++)
            if (offsets[q + 1] < offsets[q]) return fail(h, FXJPS_E_ARG, "offsets must ascend");
        if (offsets[0] != 0) return fail(h, FXJPS_E_ARG, "offsets[0] must be 0");
        if (nonneg)
            for (int64_t i = 0; i < 2 * offsets[nq]; i++)
                if (cells_xy[i] < 0) return fail(h, FXJPS_E_ARG, "negative cell");
        DevCtx& d = h->devs[0];
        HIPCHK(h, hipSetDevice(d.dev));
        const long long total = offsets[nq];
        HIPCHK(h, d.d_wp_off.ensure((size_t)nq + 1));
        HIPCHK(h, d.d_wp_len.ensure((size_t)nq));
        HIPCHK(h, d.d_wp_cells.ensure((size_t)std::max<long long>(total, 1) * 2));
        std::vector<int32_t> len((size_t)nq);
        for (int64_t q = 0; q < nq; q++) len[(size_t)q] = (int32_t)std::min<int64_t>(offsets[q + 1] - offsets[q], 0x7FFFFFFF);
        static_assert(sizeof(long long) == sizeof(int64_t), "offsets are copied as they are");
        HIPCHK(h, hipMemcpyAsync(d.d_wp_off.p, offsets, ((size_t)nq + 1) * sizeof(long long), hipMemcpyHostToDevice, d.stream));
        HIPCHK(h, hipMemcpyAsync(d.d_wp_len.p, len.data(), (size_t)nq * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
        if (total > 0) HIPCHK(h, hipMemcpyAsync(d.d_wp_cells.p, cells_xy, (size_t)total * 2 * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
        HIPCHK(h, hipStreamSynchronize(d.stream));  // (`len` is pageable host memory)
        parts.push_back(WpPart{&d, 0, nq, d.d_wp_off.p, d.d_wp_cells.p, d.d_wp_len.p, total});
        kept_at.push_back(0);
        kept_base = total;
    } else {
        for (auto& d : h->devs) {
            if (d.nq == 0) continue;
            const long long total = d.h_offsets.p[d.nq];
            parts.push_back(WpPart{&d, d.q0, d.nq, d.d_offsets.p, d.d_cells.p, d.d_len.p, total});
            kept_at.push_back(kept_base);
            kept_base += total;
        }
    }
    return FXJPS_OK;
}

}  // namespace

extern "C" {

int fxjps_version(void) { return FXJPS_VERSION; }

int fxjps_timing_size(void) { return (int)sizeof(fxjps_timing_t); }

int fxjps_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return FXJPS_E_NODEV;
    return n;
}

int fxjps_create(int backend, const int* device_ids, int n_dev, fxjps_t** out) {
    if (!out) return fail(nullptr, FXJPS_E_ARG, "out == NULL");
    *out = nullptr;
    if (backend != FXJPS_BACKEND_HIP)
        return fail(nullptr, FXJPS_E_ARG, "backend %d: only FXJPS_BACKEND_HIP exists (no CPU fallback)", backend);
    int navail = 0;
    if (hipGetDeviceCount(&navail) != hipSuccess || navail <= 0)
        return fail(nullptr, FXJPS_E_NODEV, "no HIP device visible: the planner needs an MI355X (no CPU fallback)");
    // (a device id may appear more than once: the handle then keeps several independent contexts -- streams, maps,
    // scratch, shards -- on that device; the grid reaches them by a device-to-device copy instead of the collective)
    if (n_dev < 1 || n_dev > 64 || (!device_ids && n_dev > navail))
        return fail(nullptr, FXJPS_E_ARG, "n_dev=%d but %d device(s) visible", n_dev, navail);
    if (device_ids)
        for (int r = 0; r < n_dev; r++)
            if (device_ids[r] < 0 || device_ids[r] >= navail)
                return fail(nullptr, FXJPS_E_ARG, "device id %d but %d device(s) visible", device_ids[r], navail);
    fxjps* h = new (std::nothrow) fxjps();
    if (!h) return fail(nullptr, FXJPS_E_NOMEM, "out of host memory");
    h->devs.resize(n_dev);
    for (int r = 0; r < n_dev; r++) {
        DevCtx& d = h->devs[r];
        d.dev = device_ids ? device_ids[r] : r;
        hipDeviceProp_t prop;
        hipError_t e = hipSetDevice(d.dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, d.dev);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        // (FXJPS_ONE_STREAM=1: measurement aid -- no second stream, hence no head launches, on this handle: how the runtime
        // deals its hardware queues out over the streams of many handles)
        const bool one_stream = getenv("FXJPS_ONE_STREAM") && atoi(getenv("FXJPS_ONE_STREAM")) != 0;
        if (e == hipSuccess && !one_stream) e = hipStreamCreateWithFlags(&d.stream_solo, hipStreamNonBlocking);
        if (e == hipSuccess && !one_stream) {
            // (fine-grained, mapped: the kernel counts with a system-scope atomic, the host polls)
            if (hipHostMalloc((void**)&d.solo_started, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
                *d.solo_started = 0u;
            } else {
                (void)hipGetLastError();
                d.solo_started = nullptr;  // no solo launches on this device
            }
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_solo0, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_solo1, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&d.ev0);
        if (e == hipSuccess) e = hipEventCreate(&d.ev1);
        if (e == hipSuccess) e = hipEventCreate(&d.ev_hd0);
        if (e == hipSuccess) e = hipEventCreate(&d.ev_hd1);
        if (e == hipSuccess) e = hipEventCreate(&d.ev_bt0);
        if (e == hipSuccess) e = hipEventCreate(&d.ev_bt1);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_upd, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_stage, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_ccl0, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_ccl1, hipEventDisableTiming);
        if (e != hipSuccess) {
            int rc = fail(nullptr, FXJPS_E_HIP, "device %d: %s", d.dev, hipGetErrorString(e));
            fxjps_destroy(h);
            return rc;
        }
        {
            std::vector<uint32_t> lut(11 * 256);
            for (uint32_t pd = 0; pd < 11; pd++)
                for (uint32_t m = 0; m < 256; m++) lut[pd * 256 + m] = dirlut_entry(pd, m);
            e = hipMemcpyToSymbol(HIP_SYMBOL(fx::c_dirlut), lut.data(), lut.size() * sizeof(uint32_t));
            if (e != hipSuccess) {
                int rc = fail(nullptr, FXJPS_E_HIP, "device %d: %s", d.dev, hipGetErrorString(e));
                fxjps_destroy(h);
                return rc;
            }
        }
        d.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        d.mem_total = prop.totalGlobalMem;
        {
            int khz = 0;
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d.dev) == hipSuccess && khz > 0) d.wall_khz = khz;
            (void)hipGetLastError();
        }
    }
    for (auto& a : h->devs) {
        a.share = 0;
        for (auto& b : h->devs) a.share += (a.dev == b.dev) ? 1 : 0;
    }
    *out = h;
    return FXJPS_OK;
}

int fxjps_rank_unique_id(void* out_id128) {
    if (!out_id128) return fail(nullptr, FXJPS_E_ARG, "out_id128 == NULL");
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(nullptr, FXJPS_E_COMM, "cannot load librccl.so: %s", dlerror());
    auto get = (nccl_get_uid_t)dlsym(lib, "ncclGetUniqueId");
    if (!get) return fail(nullptr, FXJPS_E_COMM, "ncclGetUniqueId not found");
    nccl_uid_t id;
    memset(&id, 0, sizeof(id));
    if (get(&id) != 0) return fail(nullptr, FXJPS_E_COMM, "ncclGetUniqueId failed");
    memcpy(out_id128, &id, sizeof(id));
    return FXJPS_OK;  // (the library stays loaded: the handle made next uses it)
}

int fxjps_rank_preflight(int device) {
    int navail = 0;
    if (hipGetDeviceCount(&navail) != hipSuccess || navail <= 0) return fail(nullptr, FXJPS_E_NODEV, "no HIP device visible");
    if (device < 0 || device >= navail) return fail(nullptr, FXJPS_E_ARG, "device id %d but %d device(s) visible", device, navail);
    void* probe = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&probe, 256);
    if (e == hipSuccess) e = hipFree(probe);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, FXJPS_E_HIP, "device %d: %s", device, hipGetErrorString(e));
    }
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(nullptr, FXJPS_E_COMM, "cannot load librccl.so: %s", dlerror());
    for (const char* sym : {"ncclGetUniqueId", "ncclCommInitRank", "ncclBroadcast", "ncclCommDestroy"})
        if (!dlsym(lib, sym)) return fail(nullptr, FXJPS_E_COMM, "%s not found in librccl.so", sym);
    return FXJPS_OK;  // (the library stays loaded: the handle made next uses it)
}

int fxjps_reserve_grid(fxjps_t* h, int32_t W, int32_t H) {
    if (!h) return FXJPS_E_ARG;
    if (W < 1 || H < 1 || W > 8190 || H > 8190) return fail(h, FXJPS_E_ARG, "grid must be 1..8190 cells a side");
    h->have_grid = false;  // (the buffers of the resident grid may have been replaced)
    h->q_results_valid = false;
    for (auto& d : h->devs) {
        int rc = alloc_grid(h, d, W, H);
        if (rc) return rc;
    }
    return FXJPS_OK;
}

int fxjps_create_rank(int device, int rank, int world, const void* id128, fxjps_t** out) {
    if (!out) return fail(nullptr, FXJPS_E_ARG, "out == NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return fail(nullptr, FXJPS_E_ARG, "rank %d of %d", rank, world);
    const int ids[1] = {device};
    fxjps_t* h = nullptr;
    int rc = fxjps_create(FXJPS_BACKEND_HIP, ids, 1, &h);
    if (rc) return rc;
    h->rank = rank;
    h->world = world;
    if (world > 1 || (getenv("FXJPS_FORCE_RCCL") && atoi(getenv("FXJPS_FORCE_RCCL")) != 0 && id128)) {
        h->rccl = load_rccl(h);
        auto init = h->rccl ? (nccl_init_rank_t)dlsym(h->rccl, "ncclCommInitRank") : nullptr;
        if (!init) {
            g_create_error = h->rccl ? "ncclCommInitRank not found" : h->err;
            fxjps_destroy(h);
            return FXJPS_E_COMM;
        }
        nccl_uid_t id;
        memcpy(&id, id128, sizeof(id));
        h->comms.assign(1, nullptr);
        (void)hipSetDevice(device);
        if (init(&h->comms[0], world, id, rank) != 0) {
            g_create_error = "ncclCommInitRank failed";
            fxjps_destroy(h);
            return FXJPS_E_COMM;
        }
    }
    *out = h;
    return FXJPS_OK;
}

int fxjps_set_grid_rank(fxjps_t* h, const uint8_t* occ, int32_t W, int32_t H) {
    if (!h) return FXJPS_E_ARG;
    if (h->world < 1) return fail(h, FXJPS_E_ARG, "fxjps_set_grid_rank on a handle that fxjps_create_rank did not make");
    if (W < 1 || H < 1 || W > 8190 || H > 8190 || (h->rank == 0 && !occ)) return fail(h, FXJPS_E_ARG, "grid must be 1..8190 cells a side (rank 0 passes it)");
    h->have_grid = false;
    h->q_results_valid = false;
    DevCtx& d0 = h->devs[0];
    int rc = alloc_grid(h, d0, W, H);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(d0.dev));
    if (h->rank == 0) HIPCHK(h, hipMemcpyAsync(d0.occ.p, occ, (size_t)W * H, hipMemcpyHostToDevice, d0.stream));
    HIPCHK(h, hipStreamSynchronize(d0.stream));
    return finish_set_grid(h, W, H);  // the broadcast (root = rank 0), then every rank derives its maps itself
}

void fxjps_destroy(fxjps_t* h) {
    if (!h) return;
    if (h->rccl) {
        auto destroy = (nccl_destroy_t)dlsym(h->rccl, "ncclCommDestroy");
        if (destroy)
            for (void* c : h->comms)
                if (c) destroy(c);
    }
    for (auto& d : h->devs) {
        if (d.dev < 0) continue;
        (void)hipSetDevice(d.dev);
        if (d.stream_solo) (void)hipStreamSynchronize(d.stream_solo);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        d.occ.release();
        d.comp.release();
        d.nb8.release();
        d.ci.release();
        d.jd.release();
        d.bm.release();
        for (int p = 0; p < 2; p++) {
            d.tables[p].release();
            d.far[p].release();
            d.wave_gen[p].release();
        }
        d.d_starts.release();
        d.d_goals.release();
        d.d_len.release();
        d.d_cells.release();
        d.d_cost.release();
        d.d_path.release();
        d.d_order.release();
        d.d_redo.release();
        d.d_offsets.release();
        d.d_next.release();
        d.d_counters.release();
        d.d_qstat.release();
        d.d_qread.release();
        d.d_raw.release();
        d.d_img.release();
        d.d_wp_in.release();
        d.d_wp_out.release();
        d.d_wp_eo.release();
        d.d_wp_nkept.release();
        d.d_wp_kept.release();
        d.d_wp_cells.release();
        d.d_wp_len.release();
        d.d_wp_off.release();
        d.d_wp_ms.release();
        d.d_wp_pdim.release();
        d.d_wp_dim.release();
        d.d_wp_prev.release();
        d.d_wp_ang.release();
        d.d_atab.release();
        d.d_upd_xy.release();
        d.d_upd_chg.release();
        d.d_owner.release();
        d.d_chgmap.release();
        d.d_chglist.release();
        d.d_chgcnt.release();
        d.d_chgovf.release();
        d.h_len.release();
        d.h_cells.release();
        d.h_cost.release();
        d.h_offsets.release();
        d.h_counters.release();
        d.h_upd_xy.release();
        d.h_path1.release();
        if (d.ev_upd) (void)hipEventDestroy(d.ev_upd);
        if (d.ev_stage) (void)hipEventDestroy(d.ev_stage);
        d.h_occ_stage.release();
        if (d.ev_ccl0) (void)hipEventDestroy(d.ev_ccl0);
        if (d.ev_ccl1) (void)hipEventDestroy(d.ev_ccl1);
        if (d.solo_started) (void)hipHostFree(d.solo_started);
        if (d.ev_solo0) (void)hipEventDestroy(d.ev_solo0);
        if (d.ev_solo1) (void)hipEventDestroy(d.ev_solo1);
        if (d.stream_solo) (void)hipStreamDestroy(d.stream_solo);

        if (d.ev0) (void)hipEventDestroy(d.ev0);
        if (d.ev1) (void)hipEventDestroy(d.ev1);
        for (hipEvent_t ev : {d.ev_hd0, d.ev_hd1, d.ev_bt0, d.ev_bt1})
            if (ev) (void)hipEventDestroy(ev);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete h;
}

const char* fxjps_last_error(fxjps_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fxjps_set_grid(fxjps_t* h, const uint8_t* occ, int32_t W, int32_t H) {
    if (!h) return FXJPS_E_ARG;
    if (!occ || W < 1 || H < 1 || W > 8190 || H > 8190) return fail(h, FXJPS_E_ARG, "grid must be 1..8190 cells a side");
    if (int rr = refuse_on_rank_handle(h, "fxjps_set_grid")) return rr;
    h->have_grid = false;
    h->q_results_valid = false;
    for (auto& d : h->devs) {
        int rc = alloc_grid(h, d, W, H);
        if (rc) return rc;
    }
    DevCtx& d0 = h->devs[0];
    HIPCHK(h, hipSetDevice(d0.dev));
    // The node's call (jps1.method once per tick on a map of ~ 150 x 110 cells): a small grid on a one-context handle goes
    // through a pinned staging buffer of the handle, and the call returns as soon as the copy and the map build are QUEUED --
    // the search of the same tick is queued right behind them, no host wait in between (round 6: 437 -> ~ 300 us for a tick
    // whose map changed; the caller's buffer is its own again when the call returns, as the ABI promises).  An error of the
    // build itself would surface at the next call that waits.  FXJPS_SETGRID_WAIT=1: wait as before (test / measurement aid).
    static const bool always_wait = getenv("FXJPS_SETGRID_WAIT") && atoi(getenv("FXJPS_SETGRID_WAIT")) != 0;
    const size_t bytes = (size_t)W * H;
    if (h->devs.size() == 1 && bytes <= ((size_t)1 << 18) && !always_wait && d0.ev_stage != nullptr) {
        if (d0.stage_pending) HIPCHK(h, hipEventSynchronize(d0.ev_stage));  // (the previous grid has left the buffer)
        d0.stage_pending = false;
        HIPCHK(h, d0.h_occ_stage.ensure(bytes));
        memcpy(d0.h_occ_stage.p, occ, bytes);
        HIPCHK(h, hipMemcpyAsync(d0.occ.p, d0.h_occ_stage.p, bytes, hipMemcpyHostToDevice, d0.stream));
        HIPCHK(h, hipEventRecord(d0.ev_stage, d0.stream));
        d0.stage_pending = true;
        return finish_set_grid(h, W, H, false);
    }
    HIPCHK(h, hipMemcpyAsync(d0.occ.p, occ, bytes, hipMemcpyHostToDevice, d0.stream));
    // (several contexts: the others copy from the first one's buffer on streams of their own.  One context: the map build
    // is queued behind the copy on the same stream, and finish_set_grid waits for both -- one host wait per call)
    if (h->devs.size() > 1) HIPCHK(h, hipStreamSynchronize(d0.stream));
    return finish_set_grid(h, W, H);
}

int fxjps_set_grid_device(fxjps_t* h, const void* d_occ, int32_t W, int32_t H) {
    if (!h) return FXJPS_E_ARG;
    if (!d_occ || W < 1 || H < 1 || W > 8190 || H > 8190) return fail(h, FXJPS_E_ARG, "grid must be 1..8190 cells a side");
    if (int rr = refuse_on_rank_handle(h, "fxjps_set_grid_device")) return rr;
    h->have_grid = false;
    h->q_results_valid = false;
    for (auto& d : h->devs) {
        int rc = alloc_grid(h, d, W, H);
        if (rc) return rc;
    }
    DevCtx& d0 = h->devs[0];
    HIPCHK(h, hipSetDevice(d0.dev));
    HIPCHK(h, hipMemcpyAsync(d0.occ.p, d_occ, (size_t)W * H, hipMemcpyDeviceToDevice, d0.stream));
    HIPCHK(h, hipStreamSynchronize(d0.stream));
    return finish_set_grid(h, W, H);
}

static int prepare_grid_impl(fxjps_t* h, const uint8_t* raw, int32_t W0, int32_t H0, int32_t ifa, int32_t variant,
                             int msg_layout, int32_t* start_xy, int32_t* goal_xy, int32_t* out_W, int32_t* out_H,
                             int32_t* out_map_d, int32_t* out_end_occu) {
    if (!h) return FXJPS_E_ARG;
    if (!raw || !start_xy || !goal_xy || W0 < 1 || H0 < 1 || ifa < 0 || ifa > 64 || (variant != 0 && variant != 1))
        return fail(h, FXJPS_E_ARG, "bad prepare_grid arguments");
    if (int rr = refuse_on_rank_handle(h, "fxjps_prepare_grid")) return rr;
    const long long sx = start_xy[0], sy = start_xy[1], gx = goal_xy[0], gy = goal_xy[1];
    // global_planner_st.py:230-235 / global_planner_ccst.py:415-420
    long long o2x = -2ll * ifa, o2y = -2ll * ifa;
    if (gx < 0 || sx < 0) o2x += std::min(gx, sx);
    if (gy < 0 || sy < 0) o2y += std::min(gy, sy);
    const long long dx = std::llabs(o2x), dy = std::llabs(o2y);
    // :246-247 / :431-432
    const long long W1 = std::max<long long>(std::max<long long>(W0, gx), sx) + dx + 4ll * ifa;
    const long long H1 = std::max<long long>(std::max<long long>(H0, gy), sy) + dy + 4ll * ifa;
    if (W1 > 8190 || H1 > 8190) return fail(h, FXJPS_E_ARG, "prepared grid %lldx%lld exceeds 8190 cells a side", W1, H1);
    h->have_grid = false;
    h->q_results_valid = false;
    for (auto& d : h->devs) {
        HIPCHK(h, hipSetDevice(d.dev));
        HIPCHK(h, d.d_raw.ensure((size_t)W0 * H0));
        int rc = alloc_grid(h, d, (int)W1, (int)H1);
        if (rc) return rc;
        // every device pads and dilates the raw grid itself: cheaper than broadcasting the larger result
        HIPCHK(h, hipMemcpyAsync(d.d_raw.p, raw, (size_t)W0 * H0, hipMemcpyHostToDevice, d.stream));
        const long long n = W1 * H1;
        hipLaunchKernelGGL(fx::k_prepare_grid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d.stream, d.d_raw.p, W0, H0,
                           (int)dx, (int)dy, ifa, variant, msg_layout, (int)W1, (int)H1, d.occ.p);
        HIPCHK(h, hipGetLastError());
        rc = derive_maps(h, d);
        if (rc) return rc;
    }
    for (auto& d : h->devs) {
        HIPCHK(h, hipSetDevice(d.dev));
        HIPCHK(h, hipStreamSynchronize(d.stream));
    }
    h->have_grid = true;
    h->maps_stale = false;
    // :266-267 (st: + map_d - 1) / :452-453 (ccst: + map_d)
    const long long sh = variant == 0 ? 1 : 0;
    long long nsx = sx + dx - sh, nsy = sy + dy - sh, ngx = gx + dx - sh, ngy = gy + dy - sh;
    if (ngx < 0 || ngy < 0 || ngx >= W1 || ngy >= H1) return fail(h, FXJPS_E_ARG, "goal outside the prepared grid");
    // :268-272 / :454-458: a goal on an obstacle moves to the nearest free cell of its row, else of its column
    DevCtx& d0 = h->devs[0];
    HIPCHK(h, hipSetDevice(d0.dev));
    std::vector<uint8_t> row((size_t)H1), col((size_t)W1);
    HIPCHK(h, hipMemcpy(row.data(), d0.occ.p + (size_t)ngx * H1, (size_t)H1, hipMemcpyDeviceToHost));
    int32_t end_occu = 0;
    if (row[(size_t)ngy]) {
        if (variant == 0) end_occu = 1;  // st:273
        long long best = -1, bd = 0;
        for (long long y = 0; y < H1; y++)
            if (!row[(size_t)y] && (best < 0 || std::llabs(y - ngy) < bd)) {  // np.argmin: first of the nearest
                best = y;
                bd = std::llabs(y - ngy);
            }
        if (best >= 0) {
            ngy = best;
        } else {
            HIPCHK(h, hipMemcpy2D(col.data(), 1, d0.occ.p + (size_t)ngy, (size_t)H1, 1, (size_t)W1, hipMemcpyDeviceToHost));
            for (long long x = 0; x < W1; x++)
                if (!col[(size_t)x] && (best < 0 || std::llabs(x - ngx) < bd)) {
                    best = x;
                    bd = std::llabs(x - ngx);
                }
            if (best < 0) return fail(h, FXJPS_E_ARG, "goal row and column are fully occupied (the reference raises here)");
            ngx = best;
        }
    }
    if (variant == 1 && ifa > 0) {
        // ccst:461-464: end_occu = (mapu[gx-ifa:gx+ifa, gy-ifa:gy+ifa] == 1).any() around the (moved) goal, with
        // numpy's slice rules: a negative bound counts from the end, everything is clipped to the array
        auto bound = [](long long v, long long n) { return v < 0 ? std::max<long long>(v + n, 0) : std::min(v, n); };
        const long long x0 = bound(ngx - ifa, W1), x1 = bound(ngx + ifa, W1), y0 = bound(ngy - ifa, H1), y1 = bound(ngy + ifa, H1);
        if (x1 > x0 && y1 > y0) {
            std::vector<uint8_t> box((size_t)((x1 - x0) * (y1 - y0)));
            HIPCHK(h, hipMemcpy2D(box.data(), (size_t)(y1 - y0), d0.occ.p + (size_t)x0 * H1 + (size_t)y0, (size_t)H1, (size_t)(y1 - y0),
                                  (size_t)(x1 - x0), hipMemcpyDeviceToHost));
            for (uint8_t v : box)
                if (v) end_occu = 1;
        }
    }
    if (out_end_occu) *out_end_occu = end_occu;
    start_xy[0] = (int32_t)nsx;
    start_xy[1] = (int32_t)nsy;
    goal_xy[0] = (int32_t)ngx;
    goal_xy[1] = (int32_t)ngy;
    if (out_W) *out_W = (int32_t)W1;
    if (out_H) *out_H = (int32_t)H1;
    if (out_map_d) {
        out_map_d[0] = (int32_t)dx;
        out_map_d[1] = (int32_t)dy;
    }
    return FXJPS_OK;
}

int fxjps_prepare_grid(fxjps_t* h, const uint8_t* raw, int32_t W0, int32_t H0, int32_t ifa, int32_t variant,
                       int32_t* start_xy, int32_t* goal_xy, int32_t* out_W, int32_t* out_H, int32_t* out_map_d,
                       int32_t* out_end_occu) {
    return prepare_grid_impl(h, raw, W0, H0, ifa, variant, 0, start_xy, goal_xy, out_W, out_H, out_map_d, out_end_occu);
}

int fxjps_prepare_occupancy_msg(fxjps_t* h, const int8_t* data, int32_t width, int32_t height, int32_t ifa, int32_t variant,
                                int32_t* start_xy, int32_t* goal_xy, int32_t* out_W, int32_t* out_H, int32_t* out_map_d,
                                int32_t* out_end_occu) {
    return prepare_grid_impl(h, reinterpret_cast<const uint8_t*>(data), width, height, ifa, variant, 1, start_xy, goal_xy,
                             out_W, out_H, out_map_d, out_end_occu);
}

int fxjps_get_grid(fxjps_t* h, uint8_t* out, int32_t* out_W, int32_t* out_H) { return fxjps_get_grid_context(h, 0, out, out_W, out_H); }

int fxjps_get_grid_context(fxjps_t* h, int32_t ctx, uint8_t* out, int32_t* out_W, int32_t* out_H) {
    if (!h) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "no grid");
    if (ctx < 0 || ctx >= (int32_t)h->devs.size()) return fail(h, FXJPS_E_ARG, "context %d of %d", (int)ctx, (int)h->devs.size());
    DevCtx& d = h->devs[(size_t)ctx];
    if (out_W) *out_W = d.W;
    if (out_H) *out_H = d.H;
    if (out) {
        // on the context's stream: it is created non-blocking, so a null-stream copy would not order behind cell
        // updates that are still queued there (fxjps_update_cells_deferred)
        HIPCHK(h, hipSetDevice(d.dev));
        HIPCHK(h, hipMemcpyAsync(out, d.occ.p, (size_t)d.W * d.H, hipMemcpyDeviceToHost, d.stream));
        HIPCHK(h, hipStreamSynchronize(d.stream));
    }
    return FXJPS_OK;
}

// ------------------------------------------------------------------ wire / on-disk adapters
namespace {
// dst[(fb ? B-1-b : b)][(fa ? A-1-a : a)][0..ch) = map(src[a][b]) for src [A][B], dst [B][A][ch]
static int transpose_map(fxjps_t* h, DevCtx& d, const uint8_t* d_src, int A, int B, int fa, int fb, int mode, int ch, uint8_t* d_dst) {
    const dim3 grid((unsigned)((B + 31) / 32), (unsigned)((A + 31) / 32)), block(32, 8);
    hipLaunchKernelGGL(fx::k_transpose_map, grid, block, 0, d.stream, d_src, A, B, fa, fb, mode, ch, d_dst);
    HIPCHK(h, hipGetLastError());
    return FXJPS_OK;
}
}  // namespace

int fxjps_publish_map(fxjps_t* h, int8_t* out_data, int32_t* out_width, int32_t* out_height) {
    if (!h) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "no grid");
    DevCtx& d = h->devs[0];
    if (out_width) *out_width = d.W;    // info.width  = len(data)      global_planner_st.py:109
    if (out_height) *out_height = d.H;  // info.height = len(data[0])   :110
    if (!out_data) return FXJPS_OK;
    HIPCHK(h, hipSetDevice(d.dev));
    const size_t n = (size_t)d.W * d.H;
    HIPCHK(h, d.d_img.ensure(n));
    int rc = transpose_map(h, d, d.occ.p, d.W, d.H, 0, 0, fx::TM_OCC_TO_MSG, 1, d.d_img.p);  // data.T, 1 -> 100   :103,115
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(out_data, d.d_img.p, n, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipStreamSynchronize(d.stream));
    return FXJPS_OK;
}

int fxjps_set_grid_image(fxjps_t* h, const uint8_t* gray, int32_t rows, int32_t cols) {
    if (!h) return FXJPS_E_ARG;
    if (!gray || rows < 1 || cols < 1 || rows > 8190 || cols > 8190) return fail(h, FXJPS_E_ARG, "image must be 1..8190 pixels a side");
    if (int rr = refuse_on_rank_handle(h, "fxjps_set_grid_image")) return rr;
    h->have_grid = false;
    h->q_results_valid = false;
    const size_t n = (size_t)rows * cols;
    for (auto& d : h->devs) {
        int rc = alloc_grid(h, d, cols, rows);  // map_pre = img[::-1].T: W = image columns, H = image rows   :182
        if (rc) return rc;
    }
    DevCtx& d0 = h->devs[0];
    HIPCHK(h, hipSetDevice(d0.dev));
    HIPCHK(h, d0.d_img.ensure(n));
    HIPCHK(h, hipMemcpyAsync(d0.d_img.p, gray, n, hipMemcpyHostToDevice, d0.stream));
    int rc = transpose_map(h, d0, d0.d_img.p, rows, cols, 1, 0, fx::TM_GRAY_TO_OCC, 1, d0.occ.p);  // > 200 white/free, else black/occupied   :179-180
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(d0.stream));
    return finish_set_grid(h, cols, rows);
}

int fxjps_snapshot_image(fxjps_t* h, uint8_t* out, int32_t channels, int32_t* out_rows, int32_t* out_cols) {
    if (!h) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "no grid");
    if (channels != 1 && channels != 3) return fail(h, FXJPS_E_ARG, "channels must be 1 (L) or 3 (RGB)");
    DevCtx& d = h->devs[0];
    if (out_rows) *out_rows = d.H;  // mapsave.T[::-1]: rows = y extent, top row = largest y   :370
    if (out_cols) *out_cols = d.W;
    if (!out) return FXJPS_OK;
    HIPCHK(h, hipSetDevice(d.dev));
    const size_t n = (size_t)d.W * d.H * channels;
    HIPCHK(h, d.d_img.ensure(n));
    int rc = transpose_map(h, d, d.occ.p, d.W, d.H, 0, 1, fx::TM_OCC_TO_GRAY, channels, d.d_img.p);  // 0 -> 255, else 0   :368-369
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(out, d.d_img.p, n, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipStreamSynchronize(d.stream));
    return FXJPS_OK;
}

namespace {
// Queue the cell updates and the rebuild of the derived maps on every device's stream (no host wait).
int update_cells_async(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n, bool derive) {
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "cell update before fxjps_set_grid");
    if (n < 0 || n > 0x7FFFFFF0ll || (n > 0 && (!xy || !val))) return fail(h, FXJPS_E_ARG, "bad update arrays");
    if (n == 0 && !(derive && h->maps_stale)) return FXJPS_OK;
    // the box of the cells of the list that lie on the grid
    int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1;
    bool have_box = false;
    {
        const DevCtx& d0 = h->devs[0];
        for (int64_t i = 0; i < n; i++) {
            const int x = xy[2 * i], y = xy[2 * i + 1];
            if (x < 0 || y < 0 || x >= d0.W || y >= d0.H) continue;
            bx0 = have_box ? std::min(bx0, x) : x;
            bx1 = have_box ? std::max(bx1, x) : x;
            by0 = have_box ? std::min(by0, y) : y;
            by1 = have_box ? std::max(by1, y) : y;
            have_box = true;
        }
    }
    static const bool partial = !(getenv("FXJPS_PARTIAL_DERIVE") && atoi(getenv("FXJPS_PARTIAL_DERIVE")) == 0);  // (0: measurement / test aid)
    // every device applies the same (small) update list; cheaper than re-broadcasting the grid
    for (auto& d : h->devs) {
        HIPCHK(h, hipSetDevice(d.dev));
        if (n > 0) {
            // The caller's arrays go through pinned staging buffers of the context: the call may return while the H2D
            // copies are still queued (fxjps_update_cells_deferred), and the caller owns its buffers again at once.
            // Before the staging buffers (and the device list the queued kernel reads) are reused, the previous
            // update's copies and kernel must be through: ev_upd.
            if (d.upd_pending) HIPCHK(h, hipEventSynchronize(d.ev_upd));
            d.upd_pending = false;
            // (coordinates and values travel as ONE copy: the values sit behind the coordinates in both buffers)
            const size_t nwords = (size_t)n * 2 + ((size_t)n + 3) / 4;
            HIPCHK(h, d.d_upd_xy.ensure(nwords));
            HIPCHK(h, d.h_upd_xy.ensure(nwords));
            memcpy(d.h_upd_xy.p, xy, (size_t)n * 2 * sizeof(int32_t));
            memcpy(d.h_upd_xy.p + (size_t)n * 2, val, (size_t)n);
            HIPCHK(h, hipMemcpyAsync(d.d_upd_xy.p, d.h_upd_xy.p, (size_t)n * 2 * sizeof(int32_t) + (size_t)n, hipMemcpyHostToDevice, d.stream));
        }
        const uint8_t* d_val = reinterpret_cast<const uint8_t*>(d.d_upd_xy.p + (size_t)n * 2);
        if (n > 0) {
            HIPCHK(h, d.d_upd_chg.ensure((size_t)n));
            if (!d.owner_ready) {  // (streaming callers only: allocated with the first update of a grid)
                HIPCHK(h, d.d_owner.ensure((size_t)d.W * d.H));
                HIPCHK(h, hipMemsetAsync(d.d_owner.p, 0xFF, (size_t)d.W * d.H * sizeof(int), d.stream));
                d.owner_ready = true;
            }
            const unsigned nbk = (unsigned)((n + 255) / 256);
            // the labels: a small update is united into them right away (the list is on the device now, not when the
            // maps are rebuilt); a large one, and every 64th small one, asks for the full relabelling
            static const long long small_max = getenv("FXJPS_CCL_SMALL") ? atoll(getenv("FXJPS_CCL_SMALL")) : 8192;  // (0: always relabel)
            const bool ccl_now = !d.ccl_full && n <= small_max && d.ccl_small < 64;
            // (measured and dropped in round 5: these four list kernels as ONE launch with barriers over its blocks -- 19 -> 15 us,
            // a 64 x 64 window update 0.091 -> 0.089 ms: a barrier across CUs costs what a launch costs)
            hipLaunchKernelGGL(fx::k_update_claim, dim3(nbk), dim3(256), 0, d.stream, d.d_owner.p, d.W, d.H, d.d_upd_xy.p, (long long)n);
            hipLaunchKernelGGL(fx::k_update_cells, dim3(nbk), dim3(256), 0, d.stream, d.occ.p, d.W, d.H, d.d_upd_xy.p, d_val,
                               (long long)n, d.d_upd_chg.p, d.d_owner.p);
            if (ccl_now) {
                // (measured and dropped in round 5: the two label kernels on the handle's second stream, beside the partial
                // rebuild that follows -- the fork / join events cost what the overlap gains: 0.091 ms either way)
                hipLaunchKernelGGL(fx::k_ccl_update, dim3(nbk), dim3(256), 0, d.stream, d.occ.p, d.W, d.H, d.d_upd_xy.p, d.d_upd_chg.p, (long long)n,
                                   d.comp.p, 0);
                hipLaunchKernelGGL(fx::k_ccl_update, dim3(nbk), dim3(256), 0, d.stream, d.occ.p, d.W, d.H, d.d_upd_xy.p, d.d_upd_chg.p, (long long)n,
                                   d.comp.p, 1);
                d.ccl_small++;
            } else {
                d.ccl_full = true;
            }
            HIPCHK(h, hipGetLastError());
            if (have_box) {  // padded coordinates: cell x sits at x + 1, its neighbours at x .. x + 2
                const int x0 = std::max(bx0, 0), x1 = std::min(bx1 + 2, d.PW - 1), y0 = std::max(by0, 0), y1 = std::min(by1 + 2, d.PH - 1);
                d.bx0 = d.dirty ? std::min(d.bx0, x0) : x0;
                d.bx1 = d.dirty ? std::max(d.bx1, x1) : x1;
                d.by0 = d.dirty ? std::min(d.by0, y0) : y0;
                d.by1 = d.dirty ? std::max(d.by1, y1) : y1;
                d.dirty = true;
            }
        }
        if (derive) {
            int rc = derive_maps(h, d, !partial);
            if (rc) {
                // The cells and the labels are already changed on the device (k_update_cells, k_ccl_update are queued), the
                // maps are not: whoever plans next must rebuild them (maps_stale), and the staging buffers stay taken until
                // what was queued has passed (ev_upd) -- the next update would otherwise overwrite a copy still in flight.
                h->maps_stale = true;
                d.dirty = true;
                if (n > 0 && hipEventRecord(d.ev_upd, d.stream) == hipSuccess) d.upd_pending = true;
                return rc;
            }
        }
        if (n > 0) {  // the list's last readers are queued: the staging buffers are free again once this event has passed
            HIPCHK(h, hipEventRecord(d.ev_upd, d.stream));
            d.upd_pending = true;
        }
    }
    h->maps_stale = !derive;
    return FXJPS_OK;
}

// Pinned staging buffer -> the caller's array.  The caller's array is as a rule freshly allocated (every page of it
// faults on the first store): beyond a few MB the copy is split over host threads (config 2: 23 MB of cells, 2.5 -> 1 ms).
static void host_copy(void* dst, const void* src, size_t bytes) {
    const size_t chunk = (size_t)2 << 20;
    const size_t nt = std::min<size_t>(8, bytes / chunk);
    if (nt < 2) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
    (void)run_side_by_side(nt, [=](size_t i) {
        const size_t o = i * per;
        if (o < bytes) memcpy((char*)dst + o, (const char*)src + o, std::min(per, bytes - o));
        return 0;
    });
}

// plan_core's results -> the caller's CSR arrays
static int emit_csr(fxjps_t* h, int64_t nq, int64_t* out_offsets, int32_t* out_cells_xy, int64_t cells_capacity, int32_t* out_len,
             double* out_cost) {
    int64_t base = 0;
    bool fits = true;
    // every context's slice of the caller's arrays: where it starts is a prefix sum over the shards' totals, the copies
    // themselves are independent -- one host thread per context when there are several
    struct Put {
        DevCtx* d;
        int64_t base, total;
        bool cells;
    };
    std::vector<Put> puts;
    for (auto& d : h->devs) {
        if (d.nq == 0) continue;
        const int64_t total = d.h_offsets.p[d.nq];
        const bool cells = out_cells_xy && base + total <= cells_capacity && total > 0;
        if (out_cells_xy && base + total > cells_capacity) fits = false;  // (out_cells_xy == NULL: sizing call, the cells stay in the handle for fxjps_last_cells())
        puts.push_back(Put{&d, base, total, cells});
        base += total;
    }
    (void)run_side_by_side(puts.size(), [&](size_t i) {
        const Put& u = puts[i];
        const DevCtx* dp = u.d;
        memcpy(out_len + dp->q0, dp->h_len.p, (size_t)dp->nq * sizeof(int32_t));
        memcpy(out_cost + dp->q0, dp->h_cost.p, (size_t)dp->nq * sizeof(double));
        for (int64_t k = 0; k < dp->nq; k++) out_offsets[dp->q0 + k] = u.base + dp->h_offsets.p[k];
        if (u.cells) host_copy(out_cells_xy + 2 * u.base, dp->h_cells.p, (size_t)u.total * 2 * sizeof(int32_t));
        return 0;
    });
    if (out_offsets) out_offsets[nq] = base;
    h->last_nq = nq;
    if (!fits) return fail(h, FXJPS_E_ARG, "out_cells_xy holds %lld pairs, batch needs %lld", (long long)cells_capacity, (long long)base);
    return FXJPS_OK;
}
}  // namespace

int fxjps_update_cells(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n) {
    if (!h) return FXJPS_E_ARG;
    h->q_results_valid = false;  // the grid changes behind the stored results
    int rc = update_cells_async(h, xy, val, n, true);
    if (rc) return rc;
    for (auto& d : h->devs) {
        HIPCHK(h, hipSetDevice(d.dev));
        HIPCHK(h, hipStreamSynchronize(d.stream));
    }
    return FXJPS_OK;
}

int fxjps_update_cells_deferred(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n) {
    if (!h) return FXJPS_E_ARG;
    h->q_results_valid = false;  // the grid changes behind the stored results
    return update_cells_async(h, xy, val, n, false);  // queued; the next planning call (or fxjps_update_cells) rebuilds the maps
}

int fxjps_set_queries(fxjps_t* h, const int32_t* starts_xy, const int32_t* goals_xy, int64_t nq, int32_t hchoice,
                      int32_t max_path_len) {
    if (!h) return FXJPS_E_ARG;
    if (nq < 0 || (nq > 0 && (!starts_xy || !goals_xy))) return fail(h, FXJPS_E_ARG, "bad query arrays");
    if (hchoice != 1 && hchoice != 2) return fail(h, FXJPS_E_ARG, "hchoice must be 1 or 2 (the reference raises TypeError, jps1.py:188)");
    if (max_path_len < 1 || max_path_len > (1 << 20)) return fail(h, FXJPS_E_ARG, "max_path_len out of range");
    h->q_starts.assign(starts_xy, starts_xy + 2 * nq);  // copied: the caller's arrays are not kept
    h->q_goals.assign(goals_xy, goals_xy + 2 * nq);
    h->q_hchoice = hchoice;
    h->q_max_len = max_path_len;
    h->q_set = true;
    h->q_results_valid = false;
    return FXJPS_OK;
}

int fxjps_replan_frame(fxjps_t* h, const int32_t* xy, const uint8_t* val, int64_t n, int64_t* out_offsets, int32_t* out_cells_xy,
                       int64_t cells_capacity, int32_t* out_len, double* out_cost, double* out_seconds_total) {
    const double t0 = now_s();
    if (!h) return FXJPS_E_ARG;
    if (!h->q_set) return fail(h, FXJPS_E_ARG, "fxjps_replan_frame before fxjps_set_queries");
    const int64_t nq = (int64_t)h->q_starts.size() / 2;
    if (nq > 0 && (!out_offsets || !out_len || !out_cost)) return fail(h, FXJPS_E_ARG, "NULL output array");
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "fxjps_replan_frame before fxjps_set_grid");
    if (n < 0 || (n > 0 && (!xy || !val))) return fail(h, FXJPS_E_ARG, "bad update arrays");
    // ---- exact reuse.  Which read-set tiles does this frame's update touch?  Derived data of a cell depends on the
    // occupancy within Chebyshev distance 1, so every changed cell marks the tiles of its 3 x 3 neighbourhood; a
    // stored result whose read set (see ReadSet in the kernels) misses all of them is what a from-scratch search
    // on the new grid would return, bit for bit, and is not searched again.
    static const bool allow_reuse = !(getenv("FXJPS_REPLAN_REUSE") && atoi(getenv("FXJPS_REPLAN_REUSE")) == 0);
    DevCtx& d0 = h->devs[0];
    unsigned long long DX[64], DY[64];  // DX[y tile]: bit per x tile, DY[x tile]: bit per y tile
    memset(DX, 0, sizeof(DX));
    memset(DY, 0, sizeof(DY));
    for (int64_t i = 0; i < n; i++) {
        const int x = xy[2 * i], y = xy[2 * i + 1];
        if (x < 0 || y < 0 || x >= d0.W || y >= d0.H) continue;  // (k_update_cells ignores it as well)
        const int tx0 = std::max(x - 1, 0) >> d0.tsh, tx1 = std::min(x + 1, d0.W - 1) >> d0.tsh;
        const int ty0 = std::max(y - 1, 0) >> d0.tsh, ty1 = std::min(y + 1, d0.H - 1) >> d0.tsh;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                DX[ty] |= 1ull << tx;
                DY[tx] |= 1ull << ty;
            }
    }
    int64_t touched = 0, tiles = (int64_t)(((d0.W - 1) >> d0.tsh) + 1) * (((d0.H - 1) >> d0.tsh) + 1);
    for (int t = 0; t < 64; t++) touched += __builtin_popcountll(DX[t]);
    // Recording read sets costs a few instructions per ray; it pays only when the next frame can reuse something.
    // A frame that touches most tiles (config 5: 10 % of all cells) leaves nothing to reuse: it runs untracked.
    const bool track = allow_reuse && 2 * touched <= tiles;
    int mode = track ? 1 : 0;
    int64_t reused = 0;
    if (track && h->q_results_valid) {
        mode = 2;
        for (auto& d : h->devs) {
            d.h_sel.assign((size_t)d.nq, 1);  // (the shards are those of the previous frame: same query set)
            const int64_t dn = d.nq;
            for (int64_t q = 0; q < dn; q++) {
                if (d.h_len.p[q] <= 0) continue;  // no path / an error code: depends on more than a read set, search again
                const unsigned long long* bx = d.h_qread.data() + (size_t)q * 128;
                const unsigned long long* by = bx + 64;
                unsigned long long hit = 0;
                for (int t = 0; t < 64; t++) hit |= (bx[t] & DX[t]) | (by[t] & DY[t]);
                if (!hit) {
                    d.h_sel[(size_t)q] = 0;
                    reused++;
                }
            }
        }
    }
    // the frame's map update is queued in front of the search on the same streams: the first host wait of the frame
    // is the one for the search results
    h->q_results_valid = false;  // (set again below, once the frame is complete)
    int rc = update_cells_async(h, xy, val, n, true);
    if (rc) {
        drain_all(h);  // the devices in front of the failing one have the update queued
        return rc;
    }
    rc = plan_core(h, h->q_starts.data(), h->q_goals.data(), nq, h->q_hchoice, h->q_max_len, mode);
    if (rc) return rc;
    h->q_results_valid = track;
    h->timing.reused = reused;
    rc = emit_csr(h, nq, out_offsets, out_cells_xy, cells_capacity, out_len, out_cost);
    h->timing.total_ms = (now_s() - t0) * 1e3;
    if (out_seconds_total) *out_seconds_total = now_s() - t0;
    return rc;
}

int fxjps_plan_batch_csr(fxjps_t* h, const int32_t* starts_xy, const int32_t* goals_xy, int64_t nq,
                         int32_t hchoice, int32_t max_path_len, int64_t* out_offsets, int32_t* out_cells_xy,
                         int64_t cells_capacity, int32_t* out_len, double* out_cost, double* out_seconds_total) {
    const double t0 = now_s();
    if (!h) return FXJPS_E_ARG;
    if (nq > 0 && (!out_offsets || !out_len || !out_cost)) return fail(h, FXJPS_E_ARG, "NULL output array");
    int rc = plan_core(h, starts_xy, goals_xy, nq, hchoice, max_path_len);
    if (rc) return rc;
    rc = emit_csr(h, nq, out_offsets, out_cells_xy, cells_capacity, out_len, out_cost);
    h->timing.total_ms = (now_s() - t0) * 1e3;
    if (out_seconds_total) *out_seconds_total = now_s() - t0;
    return rc;
}

int fxjps_last_cells(fxjps_t* h, int32_t* out_cells_xy, int64_t cells_capacity) {
    if (!h || !out_cells_xy) return FXJPS_E_ARG;
    int64_t base = 0;
    for (auto& d : h->devs) {
        if (d.nq == 0 || !d.h_offsets.p) continue;
        base += d.h_offsets.p[d.nq];
    }
    if (base > cells_capacity)
        return fail(h, FXJPS_E_ARG, "out_cells_xy holds %lld pairs, the last batch has more", (long long)cells_capacity);
    base = 0;
    struct Put {
        int32_t* dst;
        const int32_t* src;
        int64_t total;
    };
    std::vector<Put> puts;  // (several contexts: their slices are copied side by side, as in emit_csr)
    for (auto& d : h->devs) {
        if (d.nq == 0 || !d.h_offsets.p) continue;
        const int64_t total = d.h_offsets.p[d.nq];
        if (total > 0) puts.push_back(Put{out_cells_xy + 2 * base, d.h_cells.p, total});
        base += total;
    }
    (void)run_side_by_side(puts.size(), [&](size_t i) {
        host_copy(puts[i].dst, puts[i].src, (size_t)puts[i].total * 2 * sizeof(int32_t));
        return 0;
    });
    return FXJPS_OK;
}

int fxjps_plan_batch(fxjps_t* h, const int32_t* starts_xy, const int32_t* goals_xy, int64_t nq, int32_t hchoice,
                     int32_t max_path_len, int32_t* out_cells_xy, int32_t* out_len, double* out_cost,
                     double* out_seconds_total) {
    const double t0 = now_s();
    if (!h) return FXJPS_E_ARG;
    if (nq > 0 && (!out_len || !out_cost)) return fail(h, FXJPS_E_ARG, "NULL output array");
    int rc = plan_core(h, starts_xy, goals_xy, nq, hchoice, max_path_len);
    if (rc) return rc;
    for (auto& d : h->devs) {
        if (d.nq == 0) continue;
        memcpy(out_len + d.q0, d.h_len.p, (size_t)d.nq * sizeof(int32_t));
        memcpy(out_cost + d.q0, d.h_cost.p, (size_t)d.nq * sizeof(double));
        if (out_cells_xy)
            for (int64_t i = 0; i < d.nq; i++) {
                const int32_t n = d.h_len.p[i];
                if (n > 0)
                    memcpy(out_cells_xy + (size_t)(d.q0 + i) * max_path_len * 2, d.h_cells.p + 2 * d.h_offsets.p[i],
                           (size_t)n * 2 * sizeof(int32_t));
            }
    }
    h->timing.total_ms = (now_s() - t0) * 1e3;
    if (out_seconds_total) *out_seconds_total = now_s() - t0;
    return FXJPS_OK;
}

int fxjps_set_memory_share(fxjps_t* h, int32_t handles_per_device) {
    if (!h) return FXJPS_E_ARG;
    if (handles_per_device < 1 || handles_per_device > 64) return fail(h, FXJPS_E_ARG, "handles_per_device must be 1..64");
    if (h->mem_div != handles_per_device) {
        h->mem_div = handles_per_device;
        for (auto& d : h->devs) {  // the next batch sizes its pools again -- from nothing: the buffers are grow-only, and
                                   // a pool sized for the whole device would stay, and be credited to this handle's budget
            if (hipSetDevice(d.dev) == hipSuccess) {
                if (d.stream_solo) (void)hipStreamSynchronize(d.stream_solo);
                (void)hipStreamSynchronize(d.stream);
            }
            for (int p = 0; p < 2; p++) {
                d.tables[p].release();
                d.far[p].release();
                d.wave_gen[p].release();
                d.cfg[p] = ScratchCfg();
                d.pool_clean[p] = false;
            }
        }
    }
    return FXJPS_OK;
}

int fxjps_comm_info(fxjps_t* h, int32_t* out_contexts, int32_t* out_devices, int32_t* out_rccl_ranks) {
    if (!h) return FXJPS_E_ARG;
    if (out_contexts) *out_contexts = (int32_t)h->devs.size();
    if (out_devices) {
        int n = 0;
        for (size_t a = 0; a < h->devs.size(); a++) {
            bool first = true;
            for (size_t b = 0; b < a; b++)
                if (h->devs[b].dev == h->devs[a].dev) first = false;
            n += first ? 1 : 0;
        }
        *out_devices = n;
    }
    if (out_rccl_ranks) {
        *out_rccl_ranks = 0;
        if (h->rccl && !h->comms.empty() && h->comms[0]) {
            typedef int (*nccl_count_t)(void*, int*);
            auto count = (nccl_count_t)dlsym(h->rccl, "ncclCommCount");
            int n = 0;
            if (count && count(h->comms[0], &n) == 0) *out_rccl_ranks = n;
        }
    }
    return FXJPS_OK;
}

int fxjps_last_timing_device(fxjps_t* h, int32_t ctx, int32_t* out_device, int64_t* out_nq, double* out_kernel_ms, int64_t* out_waves) {
    if (!h || ctx < 0 || ctx >= (int32_t)h->devs.size()) return FXJPS_E_ARG;
    const DevCtx& d = h->devs[(size_t)ctx];
    if (out_device) *out_device = d.dev;
    if (out_nq) *out_nq = d.nq;
    if (out_kernel_ms) *out_kernel_ms = d.kernel_ms;
    if (out_waves) *out_waves = d.waves_used;
    return FXJPS_OK;
}

// ------------------------------------------------------------------ waypoint selection over a batch
int fxjps_waypoint_ccst_batch(fxjps_t* h, int64_t nq, const int64_t* offsets, const int32_t* cells_xy, double reso, const double* origin,
                              const double* pos, const double* goal, const int32_t* end_occu, double* out_wp, double* out_goal,
                              int32_t* out_n_kept, int32_t* out_kept_cells, int64_t kept_capacity) {
    if (!h) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "fxjps_waypoint_ccst_batch before fxjps_set_grid");
    if (nq < 0 || !origin || (nq > 0 && (!pos || !goal || !out_wp))) return fail(h, FXJPS_E_ARG, "bad waypoint arguments");
    if ((cells_xy != nullptr) != (offsets != nullptr)) return fail(h, FXJPS_E_ARG, "offsets and cells_xy go together");
    if (!cells_xy && nq != h->last_nq) return fail(h, FXJPS_E_ARG, "the last batch had %lld queries, not %lld", (long long)h->last_nq, (long long)nq);
    if (!cells_xy && h->last_on_host) {  // (a single call leaves its path in the pinned host buffers: handed over like a caller's CSR)
        static_assert(sizeof(long long) == sizeof(int64_t), "offsets are handed over as they are");
        offsets = reinterpret_cast<const int64_t*>(h->devs[0].h_offsets.p);
        cells_xy = h->devs[0].h_cells.p;
    }
    if (h->maps_stale) {  // (the grid is what the line test reads; deferred updates are queued on the same streams anyway)
        int rc = update_cells_async(h, nullptr, nullptr, 0, true);
        if (rc) return rc;
    }
    if (nq == 0) return FXJPS_OK;
    // the paths: the caller's CSR (device 0 takes all of them), or the last batch's, resident shard by shard
    std::vector<WpPart> parts;
    int64_t kept_base = 0;
    std::vector<int64_t> kept_at;
    {
        int rc = wp_gather_parts(h, nq, offsets, cells_xy, true, parts, kept_at, kept_base);
        if (rc) return rc;
    }
    if (out_kept_cells && kept_capacity < kept_base)
        return fail(h, FXJPS_E_ARG, "out_kept_cells holds %lld pairs, the paths have %lld", (long long)kept_capacity, (long long)kept_base);
    int rc = [&]() -> int {
    for (auto& P : parts) {  // queue every device, then collect
        DevCtx& d = *P.d;
        HIPCHK(h, hipSetDevice(d.dev));
        const size_t n = (size_t)P.n;
        HIPCHK(h, d.d_wp_in.ensure(n * 6));
        HIPCHK(h, d.d_wp_out.ensure(n * 6));
        HIPCHK(h, d.d_wp_eo.ensure(n));
        HIPCHK(h, d.d_wp_nkept.ensure(n));
        HIPCHK(h, d.d_wp_kept.ensure((size_t)std::max<long long>(P.total, 1) * 2));
        HIPCHK(h, hipMemcpyAsync(d.d_wp_in.p, pos + 3 * P.q0, n * 3 * sizeof(double), hipMemcpyHostToDevice, d.stream));
        HIPCHK(h, hipMemcpyAsync(d.d_wp_in.p + n * 3, goal + 3 * P.q0, n * 3 * sizeof(double), hipMemcpyHostToDevice, d.stream));
        if (end_occu) HIPCHK(h, hipMemcpyAsync(d.d_wp_eo.p, end_occu + P.q0, n * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
        fx::WaypointArgs A;
        A.occ = d.occ.p;
        A.W = d.W;
        A.H = d.H;
        A.cells = P.d_cells;
        A.offsets = P.d_off;
        A.len = P.d_len;
        A.nq = (long long)P.n;
        A.reso = reso;
        A.ox = origin[0];
        A.oy = origin[1];
        A.pos = d.d_wp_in.p;
        A.goal = d.d_wp_in.p + n * 3;
        A.end_occu = end_occu ? d.d_wp_eo.p : nullptr;
        A.out_wp = d.d_wp_out.p;
        A.out_goal = d.d_wp_out.p + n * 3;
        A.out_nkept = d.d_wp_nkept.p;
        A.kept = d.d_wp_kept.p;
        hipLaunchKernelGGL(fx::k_waypoint_ccst, dim3((unsigned)((P.n + 3) / 4)), dim3(256), 0, d.stream, A);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(out_wp + 3 * P.q0, d.d_wp_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
        if (out_goal) HIPCHK(h, hipMemcpyAsync(out_goal + 3 * P.q0, d.d_wp_out.p + n * 3, n * 3 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
        if (out_n_kept) HIPCHK(h, hipMemcpyAsync(out_n_kept + P.q0, d.d_wp_nkept.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
        if (out_kept_cells && P.total > 0)
            HIPCHK(h, hipMemcpyAsync(out_kept_cells + 2 * kept_at[(size_t)(&P - parts.data())], d.d_wp_kept.p, (size_t)P.total * 2 * sizeof(int32_t),
                                     hipMemcpyDeviceToHost, d.stream));
    }
    return FXJPS_OK;
    }();
    if (rc) {  // copies of the devices in front of the failing one are still queued on the caller's buffers
        drain_all(h);
        return rc;
    }
    for (auto& P : parts) {
        if (hipSetDevice(P.d->dev) != hipSuccess || hipStreamSynchronize(P.d->stream) != hipSuccess) rc = fail(h, FXJPS_E_HIP, "waypoint kernel failed");
    }
    return rc;
}

int fxjps_waypoint_st_batch(fxjps_t* h, int64_t nq, const int64_t* offsets, const int32_t* cells_xy, const int32_t* map_start, double reso,
                            const double* origin, const double* pos, const double* goal, const int32_t* end_occu, double dis_wp_tre,
                            double ang_wp_tre, const double* prev_wp, const int32_t* prev_dim, double* out_wp, int32_t* out_dim,
                            double* out_goal, double* out_ang_wp, int32_t nthreads) {
    if (!h) return FXJPS_E_ARG;
    if (nq < 0 || !origin || (nq > 0 && (!map_start || !pos || !goal || !out_wp || !out_dim || !out_goal || !out_ang_wp)))
        return fail(h, FXJPS_E_ARG, "bad waypoint arguments");
    if ((cells_xy != nullptr) != (offsets != nullptr)) return fail(h, FXJPS_E_ARG, "offsets and cells_xy go together");
    if ((prev_wp != nullptr) != (prev_dim != nullptr)) return fail(h, FXJPS_E_ARG, "prev_wp and prev_dim go together");
    if (!cells_xy && nq != h->last_nq) return fail(h, FXJPS_E_ARG, "the last batch had %lld queries, not %lld", (long long)h->last_nq, (long long)nq);
    if (nq == 0) return FXJPS_OK;
    // ---- on the device (round 6): one wavefront per path, the angles out of a table of the HOST's atan2 over the integer
    // pairs this batch can ask for.  The table is filled once per range (threads of this call) and stays on the device.
    if (!(getenv("FXJPS_WAYPOINT_ST_HOST") && atoi(getenv("FXJPS_WAYPOINT_ST_HOST")) != 0)) {  // (1: the host form, test / measurement aid)
        if (!cells_xy && h->last_on_host) {  // (a single call's path is in the pinned host buffers: handed over like a caller's CSR)
            static_assert(sizeof(long long) == sizeof(int64_t), "offsets are handed over as they are");
            offsets = reinterpret_cast<const int64_t*>(h->devs[0].h_offsets.p);
            cells_xy = h->devs[0].h_cells.p;
        }
        // the range of (cell + 1 - map_start) over the batch
        long long cx0 = 0, cx1 = 0, cy0 = 0, cy1 = 0;
        if (cells_xy) {
            if (offsets[nq] > 0) {
                cx0 = cx1 = cells_xy[0];
                cy0 = cy1 = cells_xy[1];
            }
            for (int64_t i = 0; i < offsets[nq]; i++) {
                cx0 = std::min<long long>(cx0, cells_xy[2 * i]);
                cx1 = std::max<long long>(cx1, cells_xy[2 * i]);
                cy0 = std::min<long long>(cy0, cells_xy[2 * i + 1]);
                cy1 = std::max<long long>(cy1, cells_xy[2 * i + 1]);
            }
        } else {  // (resident paths lie on the grid)
            cx1 = h->devs[0].W - 1;
            cy1 = h->devs[0].H - 1;
        }
        long long am = 0, bm = 0;
        for (int64_t q = 0; q < nq; q++) {
            const long long mx = map_start[2 * q], my = map_start[2 * q + 1];
            am = std::max(am, std::max(std::llabs(cx0 + 1 - mx), std::llabs(cx1 + 1 - mx)));
            bm = std::max(bm, std::max(std::llabs(cy0 + 1 - my), std::llabs(cy1 + 1 - my)));
        }
        constexpr long long ATAB_MAX = 1ll << 27;  // entries (1 GiB): a map_start far off the grid takes the host form below
        if ((am + 1) * (2 * bm + 1) <= ATAB_MAX && am < (1ll << 30) && bm < (1ll << 30)) {
            std::vector<WpPart> parts;
            std::vector<int64_t> kept_at;
            int64_t kept_base = 0;
            int rc = wp_gather_parts(h, nq, offsets, cells_xy, false, parts, kept_at, kept_base);
            if (rc) return rc;
            // the table: grown to the largest range seen on each device, filled by the host's libm
            std::vector<double> tab;
            int ta = -1, tb = -1;
            for (auto& P : parts) {
                DevCtx& d = *P.d;
                if (d.atab_a >= (int)am && d.atab_b >= (int)bm) continue;
                const int na = std::max<int>((int)am, d.atab_a), nb = std::max<int>((int)bm, d.atab_b);
                if ((long long)(na + 1) * (2ll * nb + 1) > ATAB_MAX) {  // (the union of two ranges may not fit: start over with this one)
                    d.atab_a = d.atab_b = -1;
                }
                const int wa = d.atab_a < 0 ? (int)am : na, wb = d.atab_b < 0 ? (int)bm : nb;
                if (ta != wa || tb != wb) {
                    ta = wa;
                    tb = wb;
                    const size_t row = (size_t)(2 * tb + 1);
                    tab.resize((size_t)(ta + 1) * row);
                    int nt = std::max(1, std::min<int>(nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency(), 64));
                    nt = (int)std::min<long long>(nt, std::max<long long>((long long)tab.size() >> 16, 1));
                    double* T = tab.data();
                    (void)run_side_by_side((size_t)nt, [&](size_t t) {
                        for (long long a = (long long)(ta + 1) * (long long)t / nt; a < (long long)(ta + 1) * (long long)(t + 1) / nt; a++)
                            for (long long b = -tb; b <= tb; b++) T[(size_t)a * row + (size_t)(b + tb)] = std::atan2((double)a, (double)b);
                        return 0;
                    });
                }
                HIPCHK(h, hipSetDevice(d.dev));
                HIPCHK(h, d.d_atab.ensure(tab.size()));
                HIPCHK(h, hipMemcpyAsync(d.d_atab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, d.stream));
                HIPCHK(h, hipStreamSynchronize(d.stream));  // (`tab` is pageable and dies with this call)
                d.atab_a = ta;
                d.atab_b = tb;
            }
            rc = [&]() -> int {
                for (auto& P : parts) {  // queue every device, then collect
                    DevCtx& d = *P.d;
                    HIPCHK(h, hipSetDevice(d.dev));
                    const size_t n = (size_t)P.n;
                    HIPCHK(h, d.d_wp_in.ensure(n * 6));
                    HIPCHK(h, d.d_wp_out.ensure(n * 6));
                    HIPCHK(h, d.d_wp_eo.ensure(n));
                    HIPCHK(h, d.d_wp_ms.ensure(n * 2));
                    HIPCHK(h, d.d_wp_prev.ensure(n * 3));
                    HIPCHK(h, d.d_wp_pdim.ensure(n));
                    HIPCHK(h, d.d_wp_dim.ensure(n));
                    HIPCHK(h, d.d_wp_ang.ensure(n));
                    HIPCHK(h, hipMemcpyAsync(d.d_wp_in.p, pos + 3 * P.q0, n * 3 * sizeof(double), hipMemcpyHostToDevice, d.stream));
                    HIPCHK(h, hipMemcpyAsync(d.d_wp_in.p + n * 3, goal + 3 * P.q0, n * 3 * sizeof(double), hipMemcpyHostToDevice, d.stream));
                    HIPCHK(h, hipMemcpyAsync(d.d_wp_ms.p, map_start + 2 * P.q0, n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
                    if (end_occu) HIPCHK(h, hipMemcpyAsync(d.d_wp_eo.p, end_occu + P.q0, n * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
                    if (prev_wp) {
                        HIPCHK(h, hipMemcpyAsync(d.d_wp_prev.p, prev_wp + 3 * P.q0, n * 3 * sizeof(double), hipMemcpyHostToDevice, d.stream));
                        HIPCHK(h, hipMemcpyAsync(d.d_wp_pdim.p, prev_dim + P.q0, n * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
                    }
                    fx::WaypointStArgs A;
                    A.cells = P.d_cells;
                    A.offsets = P.d_off;
                    A.len = P.d_len;
                    A.nq = (long long)P.n;
                    A.map_start = d.d_wp_ms.p;
                    A.reso = reso;
                    A.ox = origin[0];
                    A.oy = origin[1];
                    A.pos = d.d_wp_in.p;
                    A.goal = d.d_wp_in.p + n * 3;
                    A.end_occu = end_occu ? d.d_wp_eo.p : nullptr;
                    A.dis_wp_tre = dis_wp_tre;
                    A.ang_wp_tre = ang_wp_tre;
                    A.prev_wp = prev_wp ? d.d_wp_prev.p : nullptr;
                    A.prev_dim = prev_wp ? d.d_wp_pdim.p : nullptr;
                    A.atab = d.d_atab.p;
                    A.amax = d.atab_a;
                    A.bmax = d.atab_b;
                    A.out_wp = d.d_wp_out.p;
                    A.out_dim = d.d_wp_dim.p;
                    A.out_goal = d.d_wp_out.p + n * 3;
                    A.out_ang = d.d_wp_ang.p;
                    hipLaunchKernelGGL(fx::k_waypoint_st, dim3((unsigned)((P.n + 3) / 4)), dim3(256), 0, d.stream, A);
                    HIPCHK(h, hipGetLastError());
                    HIPCHK(h, hipMemcpyAsync(out_wp + 3 * P.q0, d.d_wp_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
                    HIPCHK(h, hipMemcpyAsync(out_goal + 3 * P.q0, d.d_wp_out.p + n * 3, n * 3 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
                    HIPCHK(h, hipMemcpyAsync(out_dim + P.q0, d.d_wp_dim.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
                    HIPCHK(h, hipMemcpyAsync(out_ang_wp + P.q0, d.d_wp_ang.p, n * sizeof(double), hipMemcpyDeviceToHost, d.stream));
                }
                return FXJPS_OK;
            }();
            if (rc) {  // copies of the devices in front of the failing one are still queued on the caller's buffers
                drain_all(h);
                return rc;
            }
            for (auto& P : parts)
                if (hipSetDevice(P.d->dev) != hipSuccess || hipStreamSynchronize(P.d->stream) != hipSuccess) rc = fail(h, FXJPS_E_HIP, "waypoint kernel failed");
            return rc;
        }
    }
    // ---- on host threads: a map_start so far off the grid that the table of angles would not fit (or FXJPS_WAYPOINT_ST_HOST=1)
    // path q: its cells and their number (the last batch's paths are in the handle's pinned host buffers, shard by shard)
    auto path_of = [&](int64_t q, const int32_t** c) -> int64_t {
        if (cells_xy) {
            *c = cells_xy + 2 * offsets[q];
            return offsets[q + 1] - offsets[q];
        }
        for (auto& d : h->devs)
            if (q >= d.q0 && q < d.q0 + d.nq) {
                const int64_t i = q - d.q0;
                *c = d.h_cells.p + 2 * d.h_offsets.p[i];
                return d.h_offsets.p[i + 1] - d.h_offsets.p[i];
            }
        *c = nullptr;
        return 0;
    };
    int nt = std::max(1, std::min<int>(nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency(), 256));
    nt = (int)std::min<int64_t>(nt, std::max<int64_t>(nq / 256, 1));
    std::vector<int> bad((size_t)nt, 0);
    auto work = [&](int t) {
        for (int64_t q = nq * t / nt; q < nq * (t + 1) / nt; q++) {
            const int32_t* c = nullptr;
            const int64_t n = path_of(q, &c);
            if (n <= 0) {  // no path: wp = global_goal    global_planner_st.py:287-290
                for (int k = 0; k < 3; k++) out_wp[3 * q + k] = out_goal[3 * q + k] = goal[3 * q + k];
                out_dim[q] = 3;
                out_ang_wp[q] = 0.0;
                continue;
            }
            const bool hp = prev_wp && (prev_dim[q] == 2 || prev_dim[q] == 3);
            if (fxjps_waypoint_st(c, (int32_t)n, map_start + 2 * q, reso, origin, pos + 3 * q, goal + 3 * q, end_occu ? end_occu[q] : 0, dis_wp_tre,
                                  ang_wp_tre, hp ? prev_wp + 3 * q : nullptr, hp ? prev_dim[q] : 0, out_wp + 3 * q, out_dim + q, out_goal + 3 * q,
                                  out_ang_wp + q) != FXJPS_OK)
                bad[(size_t)t] = 1;
        }
    };
    (void)run_side_by_side((size_t)nt, [&](size_t t) {
        work((int)t);
        return 0;
    });
    for (int b : bad)
        if (b) return fail(h, FXJPS_E_ARG, "fxjps_waypoint_st failed on a path");
    return FXJPS_OK;
}

int fxjps_last_timing(fxjps_t* h, fxjps_timing_t* out) {
    if (!h || !out) return FXJPS_E_ARG;
    *out = h->timing;
    return FXJPS_OK;
}

int fxjps_last_timing_sized(fxjps_t* h, void* out, int64_t out_size) {
    if (!h || !out || out_size < 0) return FXJPS_E_ARG;
    memcpy(out, &h->timing, (size_t)std::min<int64_t>(out_size, (int64_t)sizeof(fxjps_timing_t)));
    return FXJPS_OK;
}

int fxjps_selftest_sqrt(fxjps_t* h, uint32_t n0, uint32_t n1, double* out) {
    if (!h || !out || n1 < n0) return FXJPS_E_ARG;
    if (n1 == n0) return FXJPS_OK;
    DevCtx& d = h->devs[0];
    HIPCHK(h, hipSetDevice(d.dev));
    const size_t n = (size_t)(n1 - n0);
    double* buf = nullptr;
    HIPCHK(h, hipMalloc((void**)&buf, n * sizeof(double)));
    hipLaunchKernelGGL(fx::k_sqrt_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d.stream, n0, n1, buf);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, buf, n * sizeof(double), hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(h, FXJPS_E_HIP, "selftest: %s", hipGetErrorString(e));
    return FXJPS_OK;
}


// raw device counters of the last batch on device 0: [0] pops [1] pushes [2] refills [3] slow pops,
// [8..17] per-phase cycles in FXJPS_PROF builds (tools only)
int fxjps_debug_counters(fxjps_t* h, unsigned long long* out32) {
    if (!h || !out32) return FXJPS_E_ARG;
    DevCtx& d = h->devs[0];
    if (!d.h_counters.p) return FXJPS_E_ARG;
    memcpy(out32, d.h_counters.p, 64 * sizeof(unsigned long long));
    return FXJPS_OK;
}

// per-query diagnostics of the last batch on device 0 (FXJPS_QSTAT=1): 4 u64 per query (tools only)
int fxjps_debug_qstat(fxjps_t* h, unsigned long long* out, int64_t nq) {
    if (!h || !out) return FXJPS_E_ARG;
    DevCtx& d = h->devs[0];
    if (!d.d_qstat.p || nq > d.nq) return FXJPS_E_ARG;
    HIPCHK(h, hipSetDevice(d.dev));
    HIPCHK(h, hipMemcpy(out, d.d_qstat.p, (size_t)nq * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FXJPS_OK;
}

int fxjps_selftest_wavemin(fxjps_t* h, int32_t rounds, uint64_t seed, int64_t* mismatches) {
    if (!h || !mismatches || rounds < 1 || rounds > (1 << 16)) return FXJPS_E_ARG;
    DevCtx& d = h->devs[0];
    HIPCHK(h, hipSetDevice(d.dev));
    std::vector<uint64_t> in((size_t)rounds * 64), out((size_t)rounds * 4 + (size_t)rounds * 64);  // minima, then u32 ranks (32-bit keys, 96-bit keys)
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = 0; i < in.size(); i++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        // mix of wide-range values and many ties
        in[i] = (i % 7 == 0) ? (x & 0xFF) : ((i % 5 == 0) ? (x | 0xFFFFFFFF00000000ull) : x);
    }
    uint64_t *din = nullptr, *dout = nullptr;
    HIPCHK(h, hipMalloc((void**)&din, in.size() * 8));
    HIPCHK(h, hipMalloc((void**)&dout, out.size() * 8));
    hipError_t e = hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fx::k_selftest_wavemin, dim3(1), dim3(64), 0, d.stream, din, dout, rounds);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    if (e == hipSuccess) e = hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(din);
    (void)hipFree(dout);
    if (e != hipSuccess) return fail(h, FXJPS_E_HIP, "selftest: %s", hipGetErrorString(e));
    int64_t bad = 0;
    for (int r = 0; r < rounds; r++) {
        uint64_t m64 = ~0ull;
        uint32_t m32 = ~0u;
        for (int l = 0; l < 64; l++) {
            m64 = std::min(m64, in[(size_t)r * 64 + l]);
            m32 = std::min(m32, (uint32_t)(in[(size_t)r * 64 + l] >> 7));
        }
        if (out[(size_t)r * 4] != m64 || out[(size_t)r * 4 + 1] != m64 || out[(size_t)r * 4 + 2] != m32 || out[(size_t)r * 4 + 3] != m32) bad++;
        // wave_rank64: number of lanes with a smaller key
        const uint32_t* rk = reinterpret_cast<const uint32_t*>(out.data() + (size_t)rounds * 4) + (size_t)r * 64;
        for (int l = 0; l < 64; l++) {
            const uint32_t k = (uint32_t)(in[(size_t)r * 64 + l] >> 7);
            uint32_t c = 0;
            for (int m = 0; m < 64; m++) c += ((uint32_t)(in[(size_t)r * 64 + m] >> 7) < k) ? 1u : 0u;
            if (rk[l] != c) {
                bad++;
                break;
            }
        }
        // wave_rank96: the same for three-dword keys compared as (hi : lo : x); few distinct values per dword, so that
        // every dword decides some of the comparisons and many keys tie
        const uint32_t* rk3 = reinterpret_cast<const uint32_t*>(out.data() + (size_t)rounds * 4 + (size_t)rounds * 32) + (size_t)r * 64;
        auto k96 = [&](int l, uint32_t& hi, uint32_t& lo, uint32_t& x) {
            const uint64_t v = in[(size_t)r * 64 + l];
            hi = (uint32_t)(v >> 40) & 0x3u;
            lo = (uint32_t)v & 0x7u;
            x = (uint32_t)(v >> 9) & 0x3u;
        };
        for (int l = 0; l < 64; l++) {
            uint32_t h, lo, x, oh, ol, ox, c = 0;
            k96(l, h, lo, x);
            for (int m = 0; m < 64; m++) {
                k96(m, oh, ol, ox);
                c += (oh < h || (oh == h && (ol < lo || (ol == lo && ox < x)))) ? 1u : 0u;
            }
            if (rk3[l] != c) {
                bad++;
                break;
            }
        }
    }
    *mismatches = bad;
    return FXJPS_OK;
}

int fxjps_selftest_openlist(fxjps_t* h, int32_t banded, int32_t far_cap, int32_t near_max, double delta0, const uint64_t* keys_f,
                            const uint32_t* keys_x, int64_t nkeys, const uint32_t* step_pops, const uint32_t* step_off, int32_t nsteps,
                            uint64_t* out_f, uint32_t* out_x, uint32_t* out_slot, uint32_t* out_k, uint32_t* out_info) {
    if (!h || !keys_f || !keys_x || !step_pops || !step_off || !out_f || !out_x || !out_slot || !out_k || !out_info || nkeys < 0 ||
        nkeys > (1 << 24) || nsteps < 0 || far_cap < 64 || far_cap > (1 << 24) || (far_cap & 7) || near_max < 1 || !(delta0 > 0.0))
        return fail(h, FXJPS_E_ARG, "bad selftest arguments");
    if (step_off[0] != 0u || (int64_t)step_off[nsteps] != nkeys) return fail(h, FXJPS_E_ARG, "step offsets do not cover the keys");
    for (int32_t i = 0; i < nsteps; i++)
        if (step_off[i + 1] < step_off[i] || step_off[i + 1] - step_off[i] > 64u) return fail(h, FXJPS_E_ARG, "a step pushes at most 64 keys");
    if (banded)  // (the ring reads the cell info of an entry from the map again: the test map is 64 x 64)
        for (int64_t i = 0; i < nkeys; i++)
            if ((keys_x[i] >> 17) >= 64u || ((keys_x[i] >> 4) & 0x1FFFu) >= 64u) return fail(h, FXJPS_E_ARG, "banded selftest: cells below 64 x 64");
    DevCtx& d = h->devs[0];
    HIPCHK(h, hipSetDevice(d.dev));
    const size_t nk = (size_t)std::max<int64_t>(nkeys, 1), ns = (size_t)std::max(nsteps, 1);
    const size_t far_n = (size_t)far_cap + (size_t)far_cap / 8;
    const size_t ci_n = (size_t)66 * 128;
    // one allocation: far tier | cell info | keys f | out f | keys x | out x | out slot | step pops | step offsets | out k | info
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_far = take(far_n * sizeof(FarEnt)), o_ci = take(ci_n * 2), o_kf = take(nk * 8), o_of = take(nk * 8), o_kx = take(nk * 4),
                 o_ox = take(nk * 4), o_os = take(nk * 4), o_sp = take(ns * 4), o_so = take((ns + 1) * 4), o_ok = take(ns * 4), o_in = take(32), o_cn = take(64 * 8);
    char* buf = nullptr;
    HIPCHK(h, hipMalloc((void**)&buf, off));
    hipError_t e = hipMemsetAsync(buf, 0, off, d.stream);
    if (e == hipSuccess && nkeys > 0) e = hipMemcpyAsync(buf + o_kf, keys_f, (size_t)nkeys * 8, hipMemcpyHostToDevice, d.stream);
    if (e == hipSuccess && nkeys > 0) e = hipMemcpyAsync(buf + o_kx, keys_x, (size_t)nkeys * 4, hipMemcpyHostToDevice, d.stream);
    if (e == hipSuccess && nsteps > 0) e = hipMemcpyAsync(buf + o_sp, step_pops, (size_t)nsteps * 4, hipMemcpyHostToDevice, d.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(buf + o_so, step_off, ((size_t)nsteps + 1) * 4, hipMemcpyHostToDevice, d.stream);
    if (e == hipSuccess) {
        SearchArgs A{};
        A.G.ci = reinterpret_cast<const uint16_t*>(buf + o_ci);
        A.G.NS = 128;
        A.far = reinterpret_cast<FarEnt*>(buf + o_far);
        A.far_cap = (uint32_t)far_cap;
        A.near_max = (uint32_t)near_max;
        A.banded = banded ? 1u : 0u;
        A.out_counters = reinterpret_cast<unsigned long long*>(buf + o_cn);  // [2] far refills, [3] direct pops from the far tier
        auto kf = reinterpret_cast<const unsigned long long*>(buf + o_kf);
        auto of = reinterpret_cast<unsigned long long*>(buf + o_of);
        auto kx = reinterpret_cast<const uint32_t*>(buf + o_kx);
        auto ox = reinterpret_cast<uint32_t*>(buf + o_ox), os = reinterpret_cast<uint32_t*>(buf + o_os), ok = reinterpret_cast<uint32_t*>(buf + o_ok),
             in = reinterpret_cast<uint32_t*>(buf + o_in);
        auto sp = reinterpret_cast<const uint32_t*>(buf + o_sp), so = reinterpret_cast<const uint32_t*>(buf + o_so);
        if (banded)
            hipLaunchKernelGGL(fx::k_selftest_openlist<true>, dim3(1), dim3(64), 0, d.stream, A, kf, kx, sp, so, (uint32_t)nsteps, delta0, of, ox, os, ok, in, (uint32_t)nkeys);
        else
            hipLaunchKernelGGL(fx::k_selftest_openlist<false>, dim3(1), dim3(64), 0, d.stream, A, kf, kx, sp, so, (uint32_t)nsteps, delta0, of, ox, os, ok, in, (uint32_t)nkeys);
        e = hipGetLastError();
    }
    if (e == hipSuccess && nkeys > 0) e = hipMemcpyAsync(out_f, buf + o_of, (size_t)nkeys * 8, hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess && nkeys > 0) e = hipMemcpyAsync(out_x, buf + o_ox, (size_t)nkeys * 4, hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess && nkeys > 0) e = hipMemcpyAsync(out_slot, buf + o_os, (size_t)nkeys * 4, hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess && nsteps > 0) e = hipMemcpyAsync(out_k, buf + o_ok, (size_t)nsteps * 4, hipMemcpyDeviceToHost, d.stream);
    unsigned long long cn[4] = {0, 0, 0, 0};
    uint32_t inf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(inf, buf + o_in, sizeof(inf), hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cn, buf + o_cn, sizeof(cn), hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    (void)hipFree(buf);
    for (int i = 0; i < 8; i++) out_info[i] = inf[i];
    out_info[2] = (uint32_t)std::min<unsigned long long>(cn[2], 0xFFFFFFFFull);
    out_info[3] = (uint32_t)std::min<unsigned long long>(cn[3], 0xFFFFFFFFull);
    if (e != hipSuccess) return fail(h, FXJPS_E_HIP, "selftest: %s", hipGetErrorString(e));
    return FXJPS_OK;
}

int fxjps_debug_read_maps(fxjps_t* h, int32_t which, void* buf, int64_t capacity_bytes, int64_t* out_bytes) {
    if (!h) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "no grid");
    DevCtx& d = h->devs[0];
    HIPCHK(h, hipSetDevice(d.dev));
    if (h->maps_stale) {  // deferred cell updates: the maps follow the grid first
        int rc = update_cells_async(h, nullptr, nullptr, 0, true);
        if (rc) return rc;
    }
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = d.bm.p; bytes = (size_t)4 * d.LINES * d.WORDS * sizeof(fx::BmWord); break;  // scan words [4][LINES][WORDS] x {stop, occ}
        case 1: src = d.ci.p; bytes = (size_t)d.PW * d.NS * sizeof(uint16_t); break;               // cell infos [PW][NS] (columns >= PH unused)
        case 2: src = d.comp.p; bytes = (size_t)d.W * d.H * sizeof(int); break;                    // component forest [W][H]
        case 3: src = d.nb8.p; bytes = (size_t)d.PW * d.NS; break;                                 // neighbour bytes [PW][NS]
        case 4: src = d.bm.p + (size_t)4 * d.LINES * d.WORDS; bytes = (size_t)4 * (d.PW + d.PH - 1) * d.WORDS * sizeof(fx::BmWord); break;  // diagonal scan words
        case 5: src = d.jd.p; bytes = (size_t)d.PW * d.NS * 8 * sizeof(uint16_t); break;              // jump distances [PW][NS][8]
        default: return fail(h, FXJPS_E_ARG, "which must be 0 .. 5");
    }
    if (out_bytes) *out_bytes = (int64_t)bytes;
    if (!buf) return FXJPS_OK;
    if (capacity_bytes < (int64_t)bytes) return fail(h, FXJPS_E_ARG, "buffer holds %lld bytes, the map has %zu", (long long)capacity_bytes, bytes);
    HIPCHK(h, hipMemcpyAsync(buf, src, bytes, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipStreamSynchronize(d.stream));
    return FXJPS_OK;
}

int fxjps_debug_read_nbmask(fxjps_t* h, uint8_t* buf) {
    if (!h || !buf) return FXJPS_E_ARG;
    if (!h->have_grid) return fail(h, FXJPS_E_NOGRID, "no grid");
    DevCtx& d = h->devs[0];
    HIPCHK(h, hipSetDevice(d.dev));
    if (h->maps_stale) {  // deferred cell updates: the maps follow the grid first
        int rc = update_cells_async(h, nullptr, nullptr, 0, true);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemcpy2DAsync(buf, (size_t)d.PH, d.nb8.p, (size_t)d.NS, (size_t)d.PH, (size_t)d.PW, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(h, hipStreamSynchronize(d.stream));
    return FXJPS_OK;
}

}  // extern "C"
