/*
 * oracle/jps_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT (see jps_oracle.h).
 *
 * Plain-C restatement of the reference's grid search, scripts/jps1.py:1-246.
 * Every function cites the reference lines it follows.  The restatement keeps
 * the reference's evaluation order (Python `and`/`or` short-circuit) so that
 * the literal-mode counters equal what a counting proxy around the numpy grid
 * sees when the real jps1.py runs (tests/golden/make_golden.py records both).
 *
 * Representation differences that do not change results:
 *   - dicts came_from/gscore and set close_set (jps1.py:185-188) are dense
 *     per-cell arrays reset through a touched list;
 *   - `jumpPoint not in [j[1] for j in pqueue]` (jps1.py:224) is answered by a
 *     "seen" flag: a node that was ever pushed stays in the heap until it is
 *     popped, a popped node is closed, and closed nodes are skipped at
 *     jps1.py:218 before the test is reached;
 *   - heapq's list is a binary heap here too, but pop order does not depend on
 *     heap layout: entries are (f, (x, y)) tuples compared lexicographically,
 *     so the pop order is the total order (f, x, y).
 */
#include "jps_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double f;
    int32_t x, y;
} hent;

typedef struct {
    const uint8_t* occ;
    int32_t W, H;
    int32_t gx, gy;
    int literal;
    fxo_stats st;
    /* per-cell state */
    double* g;
    int32_t* parent; /* cell index of came_from[], -1 none */
    uint8_t* flag;   /* bit0 seen (has gscore), bit1 closed */
    int32_t* touched;
    int64_t ntouched;
    hent* heap;
    int64_t hn, hcap;
} ws_t;

static int ws_init(ws_t* w, const uint8_t* occ, int32_t W, int32_t H) {
    size_t n = (size_t)W * (size_t)H;
    memset(w, 0, sizeof(*w));
    w->occ = occ;
    w->W = W;
    w->H = H;
    w->g = (double*)malloc(n * sizeof(double));
    w->parent = (int32_t*)malloc(n * sizeof(int32_t));
    w->flag = (uint8_t*)calloc(n, 1);
    w->touched = (int32_t*)malloc(n * sizeof(int32_t));
    w->hcap = 1024;
    w->heap = (hent*)malloc((size_t)w->hcap * sizeof(hent));
    return (w->g && w->parent && w->flag && w->touched && w->heap) ? 0 : -1;
}

static void ws_free(ws_t* w) {
    free(w->g);
    free(w->parent);
    free(w->flag);
    free(w->touched);
    free(w->heap);
}

static void ws_reset(ws_t* w) {
    for (int64_t i = 0; i < w->ntouched; i++) w->flag[w->touched[i]] = 0;
    w->ntouched = 0;
    w->hn = 0;
    memset(&w->st, 0, sizeof(w->st));
}

/* ---- heap of (f, x, y): heapq.heappush / heappop, jps1.py:192,198,228 ---- */
static inline int hless(const hent* a, const hent* b) {
    if (a->f != b->f) return a->f < b->f;
    if (a->x != b->x) return a->x < b->x;
    return a->y < b->y;
}

static void hpush(ws_t* w, double f, int32_t x, int32_t y) {
    if (w->hn == w->hcap) {
        w->hcap *= 2;
        w->heap = (hent*)realloc(w->heap, (size_t)w->hcap * sizeof(hent));
    }
    hent e = {f, x, y};
    int64_t i = w->hn++;
    while (i > 0) {
        int64_t p = (i - 1) >> 1;
        if (!hless(&e, &w->heap[p])) break;
        w->heap[i] = w->heap[p];
        i = p;
    }
    w->heap[i] = e;
    w->st.pushes++;
    if (w->hn > w->st.open_peak) w->st.open_peak = w->hn;
}

static hent hpop(ws_t* w) {
    hent top = w->heap[0];
    hent e = w->heap[--w->hn];
    int64_t i = 0, n = w->hn;
    for (;;) {
        int64_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && hless(&w->heap[c + 1], &w->heap[c])) c++;
        if (!hless(&w->heap[c], &e)) break;
        w->heap[i] = w->heap[c];
        i = c;
    }
    if (n > 0) w->heap[i] = e;
    w->st.pops++;
    return top;
}

/* matrix[x][y] == 1 with one counted element read */
static inline int M1(ws_t* w, int32_t x, int32_t y) {
    w->st.cells++;
    return w->occ[(size_t)x * (size_t)w->H + (size_t)y] != 0;
}

/* jps1.py:14-31 */
static int blocked(ws_t* w, int32_t cX, int32_t cY, int32_t dX, int32_t dY) {
    if (cX + dX < 0 || cX + dX >= w->W) return 1;
    if (cY + dY < 0 || cY + dY >= w->H) return 1;
    if (dX != 0 && dY != 0) {
        if (M1(w, cX + dX, cY) && M1(w, cX, cY + dY)) return 1;
        if (M1(w, cX + dX, cY + dY)) return 1;
    } else {
        if (dX != 0) {
            if (M1(w, cX + dX, cY)) return 1;
        } else {
            if (M1(w, cX, cY + dY)) return 1;
        }
    }
    return 0;
}

/* jps1.py:34-38 */
static int dblock(ws_t* w, int32_t cX, int32_t cY, int32_t dX, int32_t dY) {
    return M1(w, cX - dX, cY) && M1(w, cX, cY - dY);
}

static inline int32_t sgn(int32_t v) { return (v > 0) - (v < 0); }

/* jps1.py:95-164.  Returns 1 and (*rx,*ry) for a jump point, 0 for None. */
static int jump(ws_t* w, int32_t cX, int32_t cY, int32_t dX, int32_t dY, int32_t* rx, int32_t* ry) {
    w->st.jumps++;
    int32_t nX = cX + dX, nY = cY + dY;
    if (blocked(w, nX, nY, 0, 0)) return 0; /* :99 */
    if (nX == w->gx && nY == w->gy) {       /* :102 */
        *rx = nX;
        *ry = nY;
        return 1;
    }
    int32_t oX = nX, oY = nY;
    if (dX != 0 && dY != 0) { /* :108-130 */
        for (;;) {
            if ((!blocked(w, oX, oY, -dX, dY) && blocked(w, oX, oY, -dX, 0)) ||
                (!blocked(w, oX, oY, dX, -dY) && blocked(w, oX, oY, 0, -dY))) {
                *rx = oX;
                *ry = oY;
                return 1;
            }
            int32_t tx, ty;
            if (jump(w, oX, oY, dX, 0, &tx, &ty) || jump(w, oX, oY, 0, dY, &tx, &ty)) { /* :116-118 */
                *rx = oX;
                *ry = oY;
                return 1;
            }
            oX += dX;
            oY += dY;
            if (blocked(w, oX, oY, 0, 0)) return 0;   /* :123 */
            if (dblock(w, oX, oY, dX, dY)) return 0;  /* :126 */
            if (oX == w->gx && oY == w->gy) {         /* :129 */
                *rx = oX;
                *ry = oY;
                return 1;
            }
        }
    } else if (dX != 0) { /* :132-146 */
        for (;;) {
            if ((!blocked(w, oX, nY, dX, 1) && blocked(w, oX, nY, 0, 1)) ||
                (!blocked(w, oX, nY, dX, -1) && blocked(w, oX, nY, 0, -1))) {
                *rx = oX;
                *ry = nY;
                return 1;
            }
            oX += dX;
            if (blocked(w, oX, nY, 0, 0)) return 0;
            if (oX == w->gx && nY == w->gy) {
                *rx = oX;
                *ry = nY;
                return 1;
            }
        }
    } else { /* :148-162 */
        for (;;) {
            if ((!blocked(w, nX, oY, 1, dY) && blocked(w, nX, oY, 1, 0)) ||
                (!blocked(w, nX, oY, -1, dY) && blocked(w, nX, oY, -1, 0))) {
                *rx = nX;
                *ry = oY;
                return 1;
            }
            oY += dY;
            if (blocked(w, nX, oY, 0, 0)) return 0;
            if (nX == w->gx && oY == w->gy) {
                *rx = nX;
                *ry = oY;
                return 1;
            }
        }
    }
}

/* jps1.py:49-93: pruned neighbour directions of (cX,cY); hasp==0 is the
 * `type(parent) != tuple` branch (start node).  Writes up to 8 (dX,dY). */
static int node_neighbours(ws_t* w, int32_t cX, int32_t cY, int hasp, int32_t pX, int32_t pY,
                           int32_t out[8][2]) {
    int n = 0;
#define ADD(ax, ay) (out[n][0] = (ax), out[n][1] = (ay), n++)
    if (!hasp) {
        static const int32_t all8[8][2] = {{-1, 0}, {0, -1}, {1, 0},  {0, 1},
                                           {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
        for (int k = 0; k < 8; k++)
            if (!blocked(w, cX, cY, all8[k][0], all8[k][1])) ADD(all8[k][0], all8[k][1]);
        return n;
    }
    int32_t dX = sgn(cX - pX), dY = sgn(cY - pY); /* direction(), :40-47 */
    if (dX != 0 && dY != 0) {                      /* :59-73 */
        if (!blocked(w, cX, cY, 0, dY)) ADD(0, dY);
        if (!blocked(w, cX, cY, dX, 0)) ADD(dX, 0);
        if ((!blocked(w, cX, cY, 0, dY) || !blocked(w, cX, cY, dX, 0)) && !blocked(w, cX, cY, dX, dY))
            ADD(dX, dY);
        if (blocked(w, cX, cY, -dX, 0) && !blocked(w, cX, cY, 0, dY)) ADD(-dX, dY);
        if (blocked(w, cX, cY, 0, -dY) && !blocked(w, cX, cY, dX, 0)) ADD(dX, -dY);
    } else if (dX == 0) { /* :76-83 */
        if (!blocked(w, cX, cY, dX, 0)) {
            if (!blocked(w, cX, cY, 0, dY)) ADD(0, dY);
            if (blocked(w, cX, cY, 1, 0)) ADD(1, dY);
            if (blocked(w, cX, cY, -1, 0)) ADD(-1, dY);
        }
    } else { /* :85-92 */
        if (!blocked(w, cX, cY, dX, 0)) {
            if (!blocked(w, cX, cY, dX, 0)) ADD(dX, 0);
            if (blocked(w, cX, cY, 0, 1)) ADD(dX, 1);
            if (blocked(w, cX, cY, 0, -1)) ADD(dX, -1);
        }
    }
#undef ADD
    return n;
}

/* jps1.py:3-12 */
static double heuristic(int32_t ax, int32_t ay, int32_t bx, int32_t by, int hchoice) {
    if (hchoice == 1) {
        double xd = fabs((double)(bx - ax)), yd = fabs((double)(by - ay));
        if (xd > yd) return 14 * yd + 10 * (xd - yd);
        return 14 * xd + 10 * (yd - xd);
    }
    int64_t dx = bx - ax, dy = by - ay;
    return sqrt((double)(dx * dx + dy * dy));
}

/* jps1.py:232-246 */
static double lenght(int32_t cx, int32_t cy, int32_t jx, int32_t jy, int hchoice) {
    if (hchoice == 1) {
        double dX = fabs((double)sgn(cx - jx)), dY = fabs((double)sgn(cy - jy));
        double lX = fabs((double)(cx - jx)), lY = fabs((double)(cy - jy));
        if (dX != 0 && dY != 0) return lX * 14;
        return (dX * lX + dY * lY) * 10;
    }
    int64_t dx = cx - jx, dy = cy - jy;
    return sqrt((double)(dx * dx + dy * dy));
}

/* jps1.py:183-230 */
static int plan_one(ws_t* w, int32_t sx, int32_t sy, int32_t gx, int32_t gy, int hchoice,
                    int32_t* out_xy, int32_t max_len, double* out_cost) {
    const int32_t W = w->W, H = w->H;
    ws_reset(w);
    *out_cost = 0.0;
    if (hchoice != 1 && hchoice != 2) return FXO_ERR_BAD_ARG; /* TypeError at :188 */
    if (sx < 0 || sy < 0 || sx >= W || sy >= H) return FXO_ERR_BAD_ARG; /* IndexError / wraparound */
    w->gx = gx;
    w->gy = gy;
    const int32_t sidx = sx * H + sy;
    w->g[sidx] = 0.0; /* gscore = {start: 0}, :187 */
    w->parent[sidx] = -1;
    w->flag[sidx] = 1;
    w->touched[w->ntouched++] = sidx;
    hpush(w, heuristic(sx, sy, gx, gy, hchoice), sx, sy); /* :192 */

    while (w->hn > 0) {
        hent cur = hpop(w); /* :198 */
        const int32_t cidx = cur.x * H + cur.y;
        if (cur.x == gx && cur.y == gy) { /* :199-208 */
            int32_t n = 0, i = cidx;
            while (i >= 0) {
                n++;
                i = w->parent[i];
            }
            *out_cost = w->g[cidx];
            if (n > max_len) return FXO_ERR_PATH_OVERFLOW;
            i = cidx;
            for (int32_t k = n - 1; k >= 0; k--) {
                out_xy[2 * k] = i / H;
                out_xy[2 * k + 1] = i % H;
                i = w->parent[i];
            }
            return n;
        }
        if (!w->literal && (w->flag[cidx] & 2)) continue; /* stale duplicate: provable no-op */
        w->flag[cidx] |= 2;                               /* close_set.add, :210 */

        /* identifySuccessors, :166-179 */
        int32_t nb[8][2];
        const int32_t pidx = w->parent[cidx];
        const int nn = node_neighbours(w, cur.x, cur.y, pidx >= 0, pidx >= 0 ? pidx / H : 0,
                                       pidx >= 0 ? pidx % H : 0, nb);
        int32_t succ[8][2];
        int ns = 0;
        for (int k = 0; k < nn; k++) {
            int32_t jx, jy;
            if (jump(w, cur.x, cur.y, nb[k][0], nb[k][1], &jx, &jy)) {
                succ[ns][0] = jx;
                succ[ns][1] = jy;
                ns++;
            }
        }
        for (int k = 0; k < ns; k++) { /* :215-228 */
            const int32_t jx = succ[k][0], jy = succ[k][1], jidx = jx * H + jy;
            if (w->flag[jidx] & 2) continue; /* :218 */
            const double tg = w->g[cidx] + lenght(cur.x, cur.y, jx, jy, hchoice); /* :221 */
            const int seen = w->flag[jidx] & 1;
            /* :223-224: tentative < gscore.get(jp, 0)  or  jp not in pqueue */
            if (tg < (seen ? w->g[jidx] : 0.0) || !seen) {
                if (!seen) {
                    w->flag[jidx] |= 1;
                    w->touched[w->ntouched++] = jidx;
                }
                w->parent[jidx] = cidx;
                w->g[jidx] = tg;
                hpush(w, tg + heuristic(jx, jy, gx, gy, hchoice), jx, jy); /* :227-228 */
            }
        }
    }
    return 0; /* :230 */
}

int fxo_version(void) { return 1; }

int fxo_plan(const uint8_t* occ, int32_t W, int32_t H, int32_t sx, int32_t sy, int32_t gx,
             int32_t gy, int32_t hchoice, int32_t literal, int32_t* out_xy, int32_t max_len,
             double* out_cost, fxo_stats* st) {
    ws_t w;
    if (ws_init(&w, occ, W, H) != 0) {
        ws_free(&w);
        return FXO_ERR_BAD_ARG;
    }
    w.literal = literal;
    int r = plan_one(&w, sx, sy, gx, gy, hchoice, out_xy, max_len, out_cost);
    if (st) *st = w.st;
    ws_free(&w);
    return r;
}

typedef struct {
    const uint8_t* occ;
    int32_t W, H;
    const int32_t *starts, *goals;
    int64_t q0, q1;
    int32_t hchoice, literal, max_len;
    int32_t* out_cells;
    int32_t* out_len;
    double* out_cost;
    fxo_stats* stats;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    ws_t w;
    if (ws_init(&w, j->occ, j->W, j->H) != 0) {
        ws_free(&w);
        for (int64_t q = j->q0; q < j->q1; q++) j->out_len[q] = FXO_ERR_BAD_ARG;
        return NULL;
    }
    w.literal = j->literal;
    for (int64_t q = j->q0; q < j->q1; q++) {
        j->out_len[q] = plan_one(&w, j->starts[2 * q], j->starts[2 * q + 1], j->goals[2 * q],
                                 j->goals[2 * q + 1], j->hchoice,
                                 j->out_cells + (size_t)q * (size_t)j->max_len * 2, j->max_len,
                                 &j->out_cost[q]);
        if (j->stats) j->stats[q] = w.st;
    }
    ws_free(&w);
    return NULL;
}

int fxo_plan_batch(const uint8_t* occ, int32_t W, int32_t H, const int32_t* starts_xy,
                   const int32_t* goals_xy, int64_t nq, int32_t hchoice, int32_t literal,
                   int32_t max_len, int32_t* out_cells_xy, int32_t* out_len, double* out_cost,
                   fxo_stats* stats, int32_t nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if ((int64_t)nthreads > nq) nthreads = nq > 0 ? (int32_t)nq : 1;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        job_t* j = &jobs[t];
        j->occ = occ;
        j->W = W;
        j->H = H;
        j->starts = starts_xy;
        j->goals = goals_xy;
        j->q0 = nq * t / nthreads;
        j->q1 = nq * (t + 1) / nthreads;
        j->hchoice = hchoice;
        j->literal = literal;
        j->max_len = max_len;
        j->out_cells = out_cells_xy;
        j->out_len = out_len;
        j->out_cost = out_cost;
        j->stats = stats;
        if (nthreads == 1)
            worker(j);
        else
            pthread_create(&th[t], NULL, worker, j);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}

/* ---- synthetic inputs (the build's own generator) ---- */
uint64_t fxo_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void fxo_synth_grid(uint8_t* occ, int32_t W, int32_t H, uint64_t seed, double p) {
    const uint64_t key = fxo_splitmix64(seed);
    const uint64_t p32 = (uint64_t)floor(p * 4294967296.0);
    const size_t n = (size_t)W * (size_t)H;
    for (size_t i = 0; i < n; i++) occ[i] = ((fxo_splitmix64(key ^ (uint64_t)i) >> 32) < p32) ? 1 : 0;
}

void fxo_synth_queries(const uint8_t* occ, int32_t W, int32_t H, uint64_t qseed, int64_t first,
                       int64_t n, int32_t* starts_xy, int32_t* goals_xy) {
    const uint64_t key = fxo_splitmix64(qseed ^ 0x51554552494553ULL); /* "QUERIES" */
    for (int64_t q = 0; q < n; q++) {
        const uint64_t i = (uint64_t)(first + q);
        int32_t pt[2][2] = {{0, 0}, {0, 0}};
        for (int k = 0; k < 2; k++) {
            for (uint64_t t = 0; t < 65536; t++) {
                const uint64_t r = fxo_splitmix64(key ^ ((i << 20) | ((uint64_t)k << 16) | t));
                const int32_t x = (int32_t)((r >> 32) % (uint64_t)W);
                const int32_t y = (int32_t)((r & 0xffffffffULL) % (uint64_t)H);
                if (occ[(size_t)x * (size_t)H + (size_t)y]) continue;
                if (k == 1 && x == pt[0][0] && y == pt[0][1]) continue;
                pt[k][0] = x;
                pt[k][1] = y;
                break;
            }
        }
        starts_xy[2 * q] = pt[0][0];
        starts_xy[2 * q + 1] = pt[0][1];
        goals_xy[2 * q] = pt[1][0];
        goals_xy[2 * q + 1] = pt[1][1];
    }
}
