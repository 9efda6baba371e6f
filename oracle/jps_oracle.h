/*
 * oracle/jps_oracle.h -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * CPU restatement (plain C) of the JPS-pruned A* grid search that
 * fuxi-planner runs once per planner tick: scripts/jps1.py:1-246 of the
 * reference, called from scripts/global_planner_st.py:285 and
 * scripts/global_planner_ccst.py:477.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped planner (libfxjps.so) never links, loads or
 * calls it.
 *
 * Parity pin: the restatement is checked cell-for-cell and cost-bit-for-bit
 * against golden vectors captured by importing the real jps1.py in the build
 * container (tests/golden/, generator tests/golden/make_golden.py).
 */
#ifndef FXO_JPS_ORACLE_H
#define FXO_JPS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-query operation counts of the LITERAL reference algorithm
 * (B_q = cells*1 + (pushes+pops)*16 bytes). */
typedef struct fxo_stats {
    int64_t cells;     /* grid element reads made by blocked()/dblock(), jps1.py:14-38 */
    int64_t pushes;    /* heapq.heappush calls, jps1.py:192,228 */
    int64_t pops;      /* heapq.heappop calls, jps1.py:198 */
    int64_t open_peak; /* max len(pqueue) */
    int64_t jumps;     /* jump() calls incl. recursive ones, jps1.py:95 */
} fxo_stats;

/* status codes in out_len: >0 path length (jump points, start and goal
 * inclusive), 0 no path, <0 error */
#define FXO_ERR_PATH_OVERFLOW (-1) /* path longer than max_len */
#define FXO_ERR_BAD_ARG (-2)       /* start out of bounds / bad hchoice */

int fxo_version(void);

/* One query.  occ[x*H + y] != 0 means "matrix[x][y] == 1" (jps1.py:20-29).
 * literal != 0: follow jps1.method step for step, including the re-expansion
 * of stale duplicate heap entries, and fill *st.  literal == 0: skip pops of
 * already-closed nodes (output-equivalent) and do not count.
 * out_xy: max_len pairs (x, y).  Returns out_len. */
int fxo_plan(const uint8_t* occ, int32_t W, int32_t H, int32_t sx, int32_t sy,
             int32_t gx, int32_t gy, int32_t hchoice, int32_t literal,
             int32_t* out_xy, int32_t max_len, double* out_cost, fxo_stats* st);

/* nq queries on nthreads host threads (contiguous static shards, each thread
 * with its own workspace).  starts/goals: nq pairs.  out_cells: nq*max_len
 * pairs.  stats may be NULL.  Returns 0. */
int fxo_plan_batch(const uint8_t* occ, int32_t W, int32_t H,
                   const int32_t* starts_xy, const int32_t* goals_xy, int64_t nq,
                   int32_t hchoice, int32_t literal, int32_t max_len,
                   int32_t* out_cells_xy, int32_t* out_len, double* out_cost,
                   fxo_stats* stats, int32_t nthreads);

/* Synthetic inputs: counter-based splitmix64. */
uint64_t fxo_splitmix64(uint64_t x);
void fxo_synth_grid(uint8_t* occ, int32_t W, int32_t H, uint64_t seed, double p);
/* i-th query draws start then goal uniformly from free cells by rejection on
 * splitmix64 counters; start != goal. */
void fxo_synth_queries(const uint8_t* occ, int32_t W, int32_t H, uint64_t qseed,
                       int64_t first, int64_t n, int32_t* starts_xy, int32_t* goals_xy);

#ifdef __cplusplus
}
#endif
#endif
