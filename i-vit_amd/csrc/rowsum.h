// rowsum.h -- torch's CPU float32 reduction order, for the literal (float view) kernels of literal.hip and ibert.hip.
#pragma once
#include "common.h"

namespace {

// float32 sum of elem(0..n-1) in the order of torch's CPU sum kernel (ATen native/cpu/SumKernel.cpp: vectorized_inner_sum ->
// row_sum -> multi_row_sum, 8-float vectors x 4 accumulator rows, 4-level cascade; oracle/ivit_oracle.c
// ivo_torch_rowsum_f32 is the checked restatement).  Whole wave calls it; the result is wave-uniform.
template <class F>
IVIT_DEV float torch_rowsum(F elem, int n, int lane)
{
    if (n < 8) {   // scalar_inner_sum: four scalar accumulators
        float fin = 0.f;
        if (lane == 0) {
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
            const int size_ilp = n >> 2;
            for (int i = 0; i < size_ilp; ++i)
                for (int k = 0; k < 4; ++k) ps[k] += elem(i * 4 + k);
            for (int i = size_ilp * 4; i < n; ++i) ps[0] += elem(i);
            for (int k = 1; k < 4; ++k) ps[0] += ps[k];
            fin = ps[0];
        }
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fin), 0));
    }
    const int vec_size = n >> 3, size_ilp = vec_size >> 2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (lane < 32) {
        int lg = 0;
        while ((1 << lg) < size_ilp) ++lg;
        const int lp = max(4, lg / 4), step = 1 << lp, mask = step - 1;
        int i = 0;
        while (i + step <= size_ilp) {
            for (int j = 0; j < step; ++j, ++i) acc0 += elem(i * 32 + lane);
            acc1 += acc0; acc0 = 0.f;
            if ((i & (mask << lp)) == 0) {
                acc2 += acc1; acc1 = 0.f;
                if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
            }
        }
        for (; i < size_ilp; ++i) acc0 += elem(i * 32 + lane);
        acc0 += acc1; acc0 += acc2; acc0 += acc3;
    }
    if (lane < 8)
        for (int i = size_ilp * 4; i < vec_size; ++i) acc0 += elem(i * 8 + lane);
    const float p1 = __shfl(acc0, (lane + 8) & 63), p2 = __shfl(acc0, (lane + 16) & 63), p3 = __shfl(acc0, (lane + 24) & 63);
    const float v = ((acc0 + p1) + p2) + p3;
    float fin = 0.f;
    for (int i = vec_size * 8; i < n; ++i) fin += elem(i);
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    return fin;
}

// The same sum when the reduced dimension is NOT the contiguous one of the tensor (a transposed view, e.g. the Swin patch embedding:
// layers_quant.py:198 hands `x.flatten(2).transpose(1, 2)` through an elementwise QuantAct, which keeps the strides, to the
// LayerNorm's x_int.mean(axis=2)).  ATen then takes vectorized_outer_sum: every output column (an index of the contiguous
// dimension, extent L) is an accumulator lane of its own.
//   columns below 32 * (L / 32) (4 vectors of 8 floats at a time): multi_row_sum -- the reduced index alone runs through the
//     4-level cascade, level_step = 2^max(4, ceil_log2(n) / 4) elements added in sequence, each finished group folded upwards;
//   the remaining L % 32 columns (vector and scalar tail): row_sum -- four interleaved partial sums (element i to partial i % 4),
//     each a cascade over n / 4 steps, the n % 4 last elements into partial 0, then ((p0 + p1) + p2) + p3.
// This is the order whenever the iteration is not split inside the contiguous extent: TensorIterator runs serially below 32768
// outputs, and with one thread (oracle/ivit_oracle.c ivo_torch_outer_rowsum_f32 is the restatement checked against torch).
// Serial; every lane that calls it gets the same value.
template <class F>
IVIT_DEV float torch_cascade_sum(F elem, int n)
{
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    const int lp = max(4, lg / 4), step = 1 << lp, mask = step - 1;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int i = 0;
    while (i + step <= n) {
        for (int j = 0; j < step; ++j, ++i) acc0 += elem(i);
        acc1 += acc0; acc0 = 0.f;
        if ((i & (mask << lp)) == 0) {
            acc2 += acc1; acc1 = 0.f;
            if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
        }
    }
    for (; i < n; ++i) acc0 += elem(i);
    acc0 += acc1; acc0 += acc2; acc0 += acc3;
    return acc0;
}

template <class F>
IVIT_DEV float torch_outer_rowsum(F elem, int n, bool tail_column)
{
    if (!tail_column) return torch_cascade_sum(elem, n);
    const int n4 = n >> 2;
    float p0 = torch_cascade_sum([&](int i) { return elem(4 * i); }, n4);
    const float p1 = torch_cascade_sum([&](int i) { return elem(4 * i + 1); }, n4);
    const float p2 = torch_cascade_sum([&](int i) { return elem(4 * i + 2); }, n4);
    const float p3 = torch_cascade_sum([&](int i) { return elem(4 * i + 3); }, n4);
    for (int i = 4 * n4; i < n; ++i) p0 += elem(i);
    return ((p0 + p1) + p2) + p3;
}

}  // namespace
