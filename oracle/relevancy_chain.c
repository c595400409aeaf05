/* oracle/relevancy_chain.c -- plain-C restatement of the single-stream relevancy chain (rules 5 + 6).
 *
 * TEST INFRASTRUCTURE ONLY: checker / CPU baseline, never linked into the product library.
 * Restates CLIP_explainability.ipynb cell 6:19-32 (batched) == ViT notebook cell 7:27-33 /
 * DETR/modules/ExplanationGenerator.py:19-24,110-118 with batch 1:
 *     R_b <- I;  for each layer l:  cam = mean_h(max(grad*attn, 0));  R_b <- R_b + cam . R_b
 * fp32 throughout, sequential summation order (heads in order, k in order).
 * Pinned by tests/test_oracle_golden.py::test_c_chain_matches_numpy_and_golden.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* attn/grad: L pointers to [B, H, N, N] fp32; R_out: [B, N, N]. Returns 0, or -1 on allocation failure. */
int oracle_self_chain_f32(const float* const* attn, const float* const* grad, int L, int B, int H, int N,
                          float* R_out) {
    const size_t nn = (size_t)N * N;
    float* cam = (float*)malloc(sizeof(float) * nn);
    float* tmp = (float*)malloc(sizeof(float) * nn);
    if (!cam || !tmp) { free(cam); free(tmp); return -1; }
    for (int b = 0; b < B; ++b) {
        float* R = R_out + (size_t)b * nn;
        memset(R, 0, sizeof(float) * nn);
        for (int i = 0; i < N; ++i) R[(size_t)i * N + i] = 1.0f;
        for (int l = 0; l < L; ++l) {
            const float* A = attn[l] + (size_t)b * H * nn;
            const float* G = grad[l] + (size_t)b * H * nn;
            for (size_t p = 0; p < nn; ++p) {
                float s = 0.0f;
                for (int h = 0; h < H; ++h) {
                    const float x = G[(size_t)h * nn + p] * A[(size_t)h * nn + p];
                    s += (x < 0.0f) ? 0.0f : x; /* clamp(min=0), NaN propagates */
                }
                cam[p] = s / (float)H;
            }
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    float acc = 0.0f;
                    for (int k = 0; k < N; ++k) acc += cam[(size_t)i * N + k] * R[(size_t)k * N + j];
                    tmp[(size_t)i * N + j] = R[(size_t)i * N + j] + acc;
                }
            memcpy(R, tmp, sizeof(float) * nn);
        }
    }
    free(cam);
    free(tmp);
    return 0;
}
