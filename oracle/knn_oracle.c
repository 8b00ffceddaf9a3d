/* CPU oracle (TEST INFRASTRUCTURE) for row a1: knn(x, k).
 *
 * Restates /root/reference/hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:4-10 with the
 * fp32 rounding sequence torch-CPU produces on this image (probed, see
 * DESIGN.md "kNN numerics"):
 *   gram(i,j)  = fmaf chain over d = 0..D-1, accumulator starts at +0
 *   sq(j)      = cascade sum of x*x: groups of 16 summed sequentially from +0,
 *                group sums accumulated sequentially, the <16 tail summed
 *                sequentially and added to the group accumulator last
 *   score(i,j) = ((-sq(j)) - (-2*gram)) - sq(i)
 * and orders candidates by (score descending, index ascending).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/build_c.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float sumsq_cascade(const float *col, int D, int stride) {
    float acc1 = 0.0f;
    int i = 0;
    while (i + 16 <= D) {
        float c = 0.0f;
        for (int j = 0; j < 16; ++j) {
            float v = col[(size_t)(i + j) * stride];
            float s = v * v;
            c = c + s;
        }
        acc1 = acc1 + c;
        i += 16;
    }
    float r = 0.0f;
    for (; i < D; ++i) {
        float v = col[(size_t)i * stride];
        float s = v * v;
        r = r + s;
    }
    return r + acc1;
}

/* x: [B, D, N] fp32 contiguous.  idx: [B, N, k] int64.  dist: [B, N, k] fp32 (may be NULL). */
int hpcs_oracle_knn(const float *x, int B, int D, int N, int k, int64_t *idx, float *dist) {
    if (B < 0 || D <= 0 || N <= 0 || k <= 0 || k > N) return -1;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * D * N;
        float *sq = (float *)malloc(sizeof(float) * (size_t)N);
        if (!sq) return -2;
        for (int j = 0; j < N; ++j) sq[j] = sumsq_cascade(xb + j, D, N);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            float *bv = (float *)malloc(sizeof(float) * (size_t)k);
            int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
            int cnt = 0;
            for (int j = 0; j < N; ++j) {
                float g = 0.0f;
                for (int d = 0; d < D; ++d) g = fmaf(xb[(size_t)d * N + i], xb[(size_t)d * N + j], g);
                float inner = -2.0f * g;
                float t = (-sq[j]) - inner;
                float s = t - sq[i];
                /* insert keeping (score desc, index asc): strict > so earlier j wins ties */
                int pos = cnt;
                while (pos > 0 && s > bv[pos - 1]) --pos;
                if (pos < k) {
                    int last = cnt < k ? cnt : k - 1;
                    for (int m = last; m > pos; --m) { bv[m] = bv[m - 1]; bi[m] = bi[m - 1]; }
                    bv[pos] = s; bi[pos] = j;
                    if (cnt < k) ++cnt;
                }
            }
            for (int m = 0; m < k; ++m) {
                idx[((size_t)b * N + i) * k + m] = bi[m];
                if (dist) dist[((size_t)b * N + i) * k + m] = bv[m];
            }
            free(bv); free(bi);
        }
        free(sq);
    }
    return 0;
}
