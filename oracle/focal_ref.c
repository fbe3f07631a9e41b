/*
 * oracle/focal_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the sigmoid focal loss used by the reference's classification loss:
 *   /root/reference/models/dino/utils.py:79-104 (sigmoid_focal_loss) with the one-hot target
 *   of /root/reference/models/dino/dino.py:517-526 expressed as a class index per row.
 * Element values follow the reference's formula in float; sums are accumulated in double so
 * the oracle is the more accurate side of any comparison.
 * Parity pin: tests/test_oracle_focal.py against tests/golden/model_units.npz (focal_*), which
 * was produced by the reference's own function.
 */
#include <math.h>
#include <stdint.h>

static float stable_bce(float x, float t) {          /* BCEWithLogits, torch's formulation */
    return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

int datr_oracle_focal_forward_f32(const float *logits, const int64_t *target, int64_t G,
                                  int64_t R, int64_t C, float alpha, float gamma, double *out_sums)
{
    for (int64_t g = 0; g < G; ++g) {
        double acc = 0.0;
        for (int64_t r = 0; r < R; ++r) {
            const int64_t row = g * R + r;
            for (int64_t c = 0; c < C; ++c) {
                const float x = logits[row * C + c];
                const float t = (c == target[row]) ? 1.f : 0.f;
                const float prob = 1.f / (1.f + expf(-x));
                const float ce = stable_bce(x, t);
                const float p_t = prob * t + (1.f - prob) * (1.f - t);
                float loss = ce * powf(1.f - p_t, gamma);
                if (alpha >= 0.f) loss = (alpha * t + (1.f - alpha) * (1.f - t)) * loss;
                acc += (double)loss;
            }
        }
        out_sums[g] = acc;
    }
    return 0;
}
