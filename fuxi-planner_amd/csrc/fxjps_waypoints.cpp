// fxjps_waypoints.cpp -- the host-only entry points of libfxjps.so (include/fxjps.h): waypoint selection after a plan.
// Plain C++ (no HIP): built into libfxjps.so by the Makefile and, with -fsanitize=address,undefined, into a checker
// library the CPU test-suite runs the golden vectors through (tests/test_sanitizers.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/fxjps.h"

extern "C" {

// ------------------------------------------------------------------ waypoint post-processing
// Host code by design: a handful of scalar float decisions on a path of a few dozen cells, whose trigonometry has to
// be the libm that CPython's math.atan2 calls.  What was O(segment x obstacles) list membership in the reference's
// map_line_col is a direct grid lookup here.

int fxjps_waypoint_st(const int32_t* cells, int32_t n, const int32_t* map_start, double reso, const double* origin, const double* pos,
                      const double* goal, int32_t end_occu, double dis_wp_tre, double ang_wp_tre, const double* prev_wp,
                      int32_t prev_dim, double* out_wp, int32_t* out_dim, double* out_goal, double* out_ang_wp) {
    if (!cells || n < 1 || !map_start || !origin || !pos || !goal || !out_wp || !out_dim || !out_goal || !out_ang_wp) return FXJPS_E_ARG;
    if (prev_wp && prev_dim != 2 && prev_dim != 3) return FXJPS_E_ARG;
    // path2 = path + (1, 1)                                                  global_planner_st.py:292
    auto px2 = [&](int k) { return cells[2 * k] + 1; };
    auto py2 = [&](int k) { return cells[2 * k + 1] + 1; };
    double wp[3] = {0.0, 0.0, 0.0};
    int dim = 0;  // 0: None
    if (prev_wp) {
        dim = prev_dim;
        for (int i = 0; i < prev_dim; i++) wp[i] = prev_wp[i];
    }
    double ang_wp = 0.0;  // :299
    const double a_goal = std::atan2((double)(px2(n - 1) - map_start[0]), (double)(py2(n - 1) - map_start[1]));
    for (int k = 1; k < n; k++) {  // :305-312
        const int dx = px2(k) - map_start[0], dy = py2(k) - map_start[1];
        const double d = std::fabs(a_goal - std::atan2((double)dx, (double)dy));
        if (d <= ang_wp && std::sqrt((double)((int64_t)dx * dx + (int64_t)dy * dy)) > 2.0) {
            wp[0] = (double)px2(k - 1) * reso + origin[0];
            wp[1] = (double)py2(k - 1) * reso + origin[1];
            dim = 2;
            break;
        }
        ang_wp = d;
    }
    for (int i = 0; i < 3; i++) out_goal[i] = goal[i];
    if (dim == 0) {  // :313-314
        for (int i = 0; i < 3; i++) wp[i] = goal[i];
        dim = 3;
    }
    const double ex = wp[0] - pos[0], ey = wp[1] - pos[1];
    const double uav2next_wp = std::sqrt(ex * ex + ey * ey);  // :315
    if (end_occu == 1) {  // :317-320
        for (int i = 0; i < 3; i++) wp[i] = out_goal[i] = pos[i];
        dim = 3;
    } else if (!(n > 2 && (uav2next_wp > dis_wp_tre || (ang_wp > ang_wp_tre && ang_wp < M_PI * 0.5)))) {  // :321-322
        for (int i = 0; i < 3; i++) wp[i] = goal[i];
        dim = 3;
    }
    for (int i = 0; i < 3; i++) out_wp[i] = i < dim ? wp[i] : 0.0;
    *out_dim = dim;
    *out_ang_wp = ang_wp;
    return FXJPS_OK;
}

namespace {
// global_planner_ccst.py:258-283 on the sub-map [x0, x1) x [y0, y1) of occ; a is the end point named p1 there, b p2.
static bool line_is_free(const uint8_t* occ, int32_t W, int32_t H, const int32_t* a, const int32_t* b) {
    const int x0 = std::min(a[0], b[0]), x1 = std::max(a[0], b[0]), y0 = std::min(a[1], b[1]), y1 = std::max(a[1], b[1]);
    const int cx1 = std::min(x1, (int)W), cy1 = std::min(y1, (int)H);  // slices clip at the array bounds
    bool any = false;
    for (int x = std::max(x0, 0); x < cx1 && !any; x++)
        for (int y = std::max(y0, 0); y < cy1; y++)
            if (occ[(size_t)x * H + y] == 1) {
                any = true;
                break;
            }
    if (!any) return true;  // no obstacle in the sub-map
    // relative to p0 = (x0, y0); p1 is the end with the smaller x (ties keep a)
    double p1x = (double)(a[0] - x0), p1y = (double)(a[1] - y0), p2x = (double)(b[0] - x0), p2y = (double)(b[1] - y0);
    if (p2x < p1x) {
        std::swap(p1x, p2x);
        std::swap(p1y, p2y);
    }
    const double slope = (p2y - p1y) / (p2x - p1x);
    for (double x = p1x + 1.0; x < p2x; x += 1.0) {  // np.arange(p1[0] + 1, p2[0], 1)
        const int cx = (int)x, cy = (int)std::nearbyint(slope * x) + (int)p1y;
        if (cx >= 0 && cx < x1 - x0 && cy >= 0 && cy < y1 - y0) {
            const int gx = x0 + cx, gy = y0 + cy;
            if (gx < W && gy < H && occ[(size_t)gx * H + gy] == 1) return false;  // collide
        }
    }
    return true;
}
}  // namespace

int fxjps_waypoint_ccst(const int32_t* cells, int32_t n, const uint8_t* occ, int32_t W, int32_t H, double reso, const double* origin,
                        const double* pos, const double* goal, int32_t end_occu, double* out_wp, double* out_goal,
                        int32_t* kept_cells, int32_t* n_kept) {
    if (!cells || n < 1 || !occ || W < 1 || H < 1 || !origin || !pos || !goal || !out_wp) return FXJPS_E_ARG;
    for (int i = 0; i < n; i++)
        if (cells[2 * i] < 0 || cells[2 * i + 1] < 0) return FXJPS_E_ARG;
    // path2 = path + (1, 0); path2_c = path; path3 = path2 * reso + origin, z = 0       ccst:487-495
    std::vector<int32_t> c2(cells, cells + 2 * (size_t)n);
    std::vector<double> p4((size_t)n * 3);
    for (int i = 0; i < n; i++) {
        p4[3 * i] = (double)(cells[2 * i] + 1) * reso + origin[0];
        p4[3 * i + 1] = (double)cells[2 * i + 1] * reso + origin[1];
        p4[3 * i + 2] = 0.0;
    }
    int m = n;
    if (n > 2) {  // :507-513: drop the points (but the first) closer than 1.5 to the vehicle
        int w = 1;
        for (int i = 1; i < n; i++) {
            const double dx = p4[3 * i] - pos[0], dy = p4[3 * i + 1] - pos[1], dz = p4[3 * i + 2] - pos[2];
            if (std::sqrt(dx * dx + dy * dy + dz * dz) < 1.5) continue;
            if (w != i) {
                for (int k = 0; k < 3; k++) p4[3 * w + k] = p4[3 * i + k];
                c2[2 * w] = c2[2 * i];
                c2[2 * w + 1] = c2[2 * i + 1];
            }
            w++;
        }
        m = w;
    }
    int ii = 1;  // :515-521: drop a point when the straight line between its neighbours is free
    while (ii < m - 1) {
        if (line_is_free(occ, W, H, &c2[2 * (ii - 1)], &c2[2 * (ii + 1)])) {
            for (int k = ii; k < m - 1; k++) {
                for (int q = 0; q < 3; q++) p4[3 * k + q] = p4[3 * (k + 1) + q];
                c2[2 * k] = c2[2 * (k + 1)];
                c2[2 * k + 1] = c2[2 * (k + 1) + 1];
            }
            m--;
        } else {
            ii++;
        }
    }
    if (m > 2) {  // :523-526
        for (int k = 0; k < 3; k++) out_wp[k] = (p4[3 + k] * 1.4 + p4[6 + k] * 0.6) / 2;
    } else {
        for (int k = 0; k < 3; k++) out_wp[k] = goal[k];
    }
    if (out_goal)
        for (int k = 0; k < 3; k++) out_goal[k] = goal[k];
    if (end_occu == 1) {  // :541-544: the goal region is occupied: hold position, the vehicle position becomes the goal
        for (int k = 0; k < 3; k++) {
            out_wp[k] = pos[k];
            if (out_goal) out_goal[k] = pos[k];
        }
    }
    if (kept_cells)
        for (int i = 0; i < 2 * m; i++) kept_cells[i] = c2[i];
    if (n_kept) *n_kept = m;
    return FXJPS_OK;
}

}  // extern "C"
