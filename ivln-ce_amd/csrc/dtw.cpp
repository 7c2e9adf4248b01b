// Host-side windowed DTW (step pattern "symmetric1") for the t-nDTW metric: replaces
// dtw.dtw(ap, gtp, step_pattern="symmetric1", window_type=window_align_func, ...).distance of
// dtw-python 1.3.0 (un-vendored; call site habitat_extensions/tour_ndtw.py:118-124).
//   D[i][j] = |a_i - b_j| + min(D[i-1][j-1], D[i-1][j], D[i][j-1]) over cells where window != 0
// distance = D[n-1][m-1] (un-normalised); +inf when no admissible warping path exists.
#include <math.h>
#include <stdint.h>
#include <vector>
#include <limits>
#include "../../include/ivln_hip.h"

extern "C" int ivln_dtw_symmetric1(const double* a, int n, const double* b, int m, int dim,
                                   const uint8_t* window, double* distance_out) {
    if (!a || !b || !distance_out || n <= 0 || m <= 0 || dim <= 0) return IVLN_E_INVALID;
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> prev(m, INF), cur(m, INF);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < m; ++j) {
            if (window && !window[(int64_t)i * m + j]) {
                cur[j] = INF;
                continue;
            }
            double d = 0.0;
            for (int k = 0; k < dim; ++k) {
                double t = a[(int64_t)i * dim + k] - b[(int64_t)j * dim + k];
                d += t * t;
            }
            d = sqrt(d);
            double best;
            if (i == 0 && j == 0) best = 0.0;
            else {
                best = INF;
                if (i > 0 && j > 0) best = fmin(best, prev[j - 1]);
                if (i > 0) best = fmin(best, prev[j]);
                if (j > 0) best = fmin(best, cur[j - 1]);
            }
            cur[j] = d + best;
        }
        prev.swap(cur);
    }
    *distance_out = prev[m - 1];
    return IVLN_OK;
}
