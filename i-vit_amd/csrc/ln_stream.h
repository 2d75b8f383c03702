// ln_stream.h -- layernorm_i8_stream_kernel: the int8 I-LayerNorm as a STREAMING kernel (round 4).  Included by rowops.hip
// inside its anonymous namespace (uses LnArgs, ln_mean, NT, WPB, v2f, dyadic_mult, sx8, pack4 from there).
// Reference: /root/reference/models/quantization_utils/ivit_modules.py:30-65 (IVITIntLayerNorm), quant_utils.py:220-230.
//
// Why another form.  layernorm_i8_v2_kernel ran at 0.34 of the HBM peak (27.5 us for 75 MB at the headline shape) and its
// phases added up (read 10 us + arithmetic 12 us + stores 6 us): the whole problem is 151 KB in + 151 KB out per CU, every
// wave took one group of rows, so all resident waves loaded, then all computed, then all stored -- 1.5 generations of
// lock-stepped waves, nothing in steady state.  Here:
//   * one persistent set of workgroups (<= 2 per CU, forced by the LDS request), every wave owns a contiguous run of row
//     groups and walks it through a RING of NG register buffers: the loads of group j + NG are issued right after group j
//     is consumed, so NG groups (NG x 64 x 16 x NC bytes per wave) are always in flight under the arithmetic of the
//     current one, and the stores of group j drain under group j + 1.  Plain C++ loads: the compiler's own counted
//     s_waitcnt vmcnt(N) ties each buffer to its consumer (checked in the ISA: csrc/check_resources.py `ln_stream`);
//   * 16 bytes per lane and instruction on both sides.  LPR lanes share a row (C = LPR x 16 x NC bytes), a group is
//     64 / LPR rows; lane (lr, lq) owns the chunks {i * LPR + lq} of row lr: every load instruction reads LPR x 16
//     contiguous bytes per row (256 B at C = 768), every store writes whole 64-byte segments of the GEMM block layout
//     (4 consecutive rows x 64 B = 256 contiguous bytes per column block) or 256 contiguous bytes of a row;
//   * row sums: v_dot4 per dword, then log2(LPR) DPP steps (quad_perm, row_half_mirror, row_mirror) -- every lane of a row
//     ends with the totals and evaluates its row's statistics itself (no broadcast); the ten Newton steps use
//     v_rcp_f32 + one exact remainder fix-up instead of IEEE divisions (proof below);
//   * per-channel constants (bias, the float32 bracket of the requantiser) once per WORKGROUP into LDS, stored chunk-
//     transposed so that the 16 lanes of a row read 16 consecutive float4 (conflict-free ds_read_b128); their global loads
//     are issued BEFORE the first row loads (vmcnt completes in order: behind them they would wait for the whole ring).
// Arithmetic identical to layernorm_i8_kernel / _v2 (same certificate, same literal fallback per 16-byte chunk, same COMPAT
// remap and tie handling); parity: tests/test_gpu_ops.py test_layernorm_* (all forms), test_producers_write_block_layout.

// all-reduce over the LPR lanes that share a row (LPR <= 16: inside one DPP row)
template <int LPR>
IVIT_DEV int ln_row_allreduce(int v)
{
    if constexpr (LPR >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    if constexpr (LPR >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    if constexpr (LPR >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    if constexpr (LPR >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return v;
}

// (ln_newton10 / ln_std10 / ln_hfactor_small: rowops.hip, beside ln_factor -- the half-wave kernel uses them too)

// torch's float32 row sum (rowops.hip torch_rowsum_phi, inner-dimension form) over a row addressed through `get(i)`
template <typename GET>
IVIT_DEV float torch_rowsum_get(GET get, int C, int lane)
{
    const int vec_size = C >> 3, size_ilp = vec_size >> 2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (lane < 32) {
        int lg = 0;
        while ((1 << lg) < size_ilp) ++lg;
        const int lp = max(4, lg / 4), step = 1 << lp, mask = step - 1;
        int i = 0;
        while (i + step <= size_ilp) {
            for (int j = 0; j < step; ++j, ++i) acc0 += get(i * 32 + lane);
            acc1 += acc0; acc0 = 0.f;
            if ((i & (mask << lp)) == 0) {
                acc2 += acc1; acc1 = 0.f;
                if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
            }
        }
        for (; i < size_ilp; ++i) acc0 += get(i * 32 + lane);
        acc0 += acc1; acc0 += acc2; acc0 += acc3;
    }
    if (lane < 8)
        for (int i = size_ilp * 4; i < vec_size; ++i) acc0 += get(i * 8 + lane);
    const float p1 = __shfl(acc0, (lane + 8) & 63), p2 = __shfl(acc0, (lane + 16) & 63), p3 = __shfl(acc0, (lane + 24) & 63);
    const float v = ((acc0 + p1) + p2) + p3;
    float fin = 0.f;
    for (int i = vec_size * 8; i < C; ++i) fin += get(i);
#pragma unroll
    for (int l = 0; l < 8; ++l) fin += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    return fin;
}


typedef unsigned ln_v4u __attribute__((ext_vector_type(4)));
constexpr unsigned LN_OOR = 0x80000000u;       // a buffer offset beyond any num_records: the load returns 0, the store is dropped

template <int LPR, int NC, int NG, bool COMPAT, int OCC = 2>
__global__ __launch_bounds__(NT, OCC) void layernorm_i8_stream_kernel(LnArgs a)
{
    static_assert(LPR == 4 || LPR == 8 || LPR == 16, "LPR");
    constexpr int RPG = 64 / LPR;      // rows per group
    constexpr int NQ = LPR * NC;       // 16-byte chunks per row
    constexpr int C = NQ * 16;
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];   // [bias | lo | hi][4 dwords of a chunk][NQ] float4
    __shared__ unsigned char s_remap[COMPAT ? 256 : 4];
    __shared__ float s_phi[COMPAT ? 256 : 1];
    __shared__ __attribute__((aligned(16))) unsigned char s_row[COMPAT ? WPB : 1][COMPAT ? C : 16];   // a tie row, per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & (LPR - 1), lr = lane / LPR;
    const int abl = IVIT_LAB ? a.abl : 0;
#if IVIT_LAB
    unsigned long long stamp[8] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#define LN_STAMP(i) do { if (a.stamps) stamp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LN_STAMP(i) do { } while (0)
#endif
    // Rows go through BUFFER loads / stores: a lane whose row does not exist (beyond the matrix, or a ring slot beyond the wave's
    // last group) gets the offset LN_OOR, the hardware's range check then drops the access.  Every load and store of the loop is
    // therefore unconditional, the number of vector-memory operations between a ring slot's load and its use is the same on
    // every path, and the compiler's s_waitcnt vmcnt(N) stays COUNTED (with branches around them it fell back to vmcnt(0)
    // at the loop head: the whole ring drained before every round).
    const unsigned x_bytes = (unsigned)((int64_t)(a.rows - 1) * a.ldx + C);
    const unsigned o_bytes = a.out_blocks ? (unsigned)(((a.rows + 15) >> 4) << 4) * (unsigned)C : (unsigned)((int64_t)(a.rows - 1) * a.ldo + C);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sln = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.s_ln), 0, C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.m), 0, C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.e), 0, C * 4, 0x00020000);

    // (1) raw per-channel constants of this thread's channels: issued first, they return first (vmcnt completes in order)
    constexpr int NCH = (C + NT - 1) / NT;
    uint32_t rm[NCH];
    int32_t re[NCH];
    float rs[NCH], rb[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = min(tid + j * NT, C - 1);
        rm[j] = a.m[c];
        re[j] = a.e[c];
        rs[j] = a.s_ln[c];
        rb[j] = a.bias_int[c];
    }
    uint32_t r_remap = 0;
    float r_phi = 0.f;
    if constexpr (COMPAT) {
        r_remap = (unsigned char)a.remap[tid];   // NT == 256
        r_phi = a.phi[tid];
    }
    __builtin_amdgcn_sched_barrier(0);

    // (2) this wave's contiguous run of row groups
    const int n_groups = (a.rows + RPG - 1) / RPG;
    const int n_waves = gridDim.x * WPB, wave_id = blockIdx.x * WPB + wave;
    const int gq = n_groups / n_waves, grm = n_groups - gq * n_waves;
    const int g_begin = wave_id * gq + min(wave_id, grm), g_cnt = gq + (wave_id < grm ? 1 : 0);
    const int row_l0 = g_begin * RPG + lr;          // this lane's row in group 0
    constexpr int NB = NG > 0 ? NG : 2;     // register buffers: NG groups per round, or the two of the double-buffered loop (NG = 0)
    ln_v4u w[NB][NC];
    auto load_group = [&](int gi, ln_v4u (&dst)[NC]) {
        const int row = row_l0 + gi * RPG;
        const unsigned off = (gi < g_cnt && row < a.rows) ? (unsigned)row * (unsigned)a.ldx + (unsigned)(lq * 16) : LN_OOR;
#pragma unroll
        for (int i = 0; i < NC; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned)(i * LPR * 16), 0, 0);
    };
    // (3) the constants table; channels beyond C are clamped duplicates of the last one.  The rows are requested only AFTER the
    // table stands: requested first (round-4 timeline, profiles/r04c_*), the 144 KB per CU of row loads filled the CU's
    // vector-memory queue and the table loads of late-starting waves came back 4 us (median; up to 8.5 us) into the kernel
    float* t_bias = lds_tab;
    float* t_lo = lds_tab + C;
    float* t_hi = lds_tab + 2 * C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = min(tid + j * NT, C - 1);
        const double M = dyadic_mult(rm[j], re[j]);
        const double lod = M * (1.0 - 1.25 / 4194304.0), hid = M * (1.0 + 1.25 / 4194304.0);
        float lf = (float)lod, hf = (float)hid;
        if ((double)lf > lod) lf = __int_as_float(__float_as_int(lf) - 1);   // largest float32 <= lod (lod > 0)
        if ((double)hf < hid) hf = __int_as_float(__float_as_int(hf) + 1);   // smallest float32 >= hid
        const float sl = rs[j];
        const bool ok = fabsf(sl) >= 1e-30f && fabsf(sl) <= 1e30f && lod > 1e-35 && hid < 1e30;   // see layernorm_i8_kernel
        const int idx = ((((c >> 2) & 3) * NQ + (c >> 4)) << 2) + (c & 3);
        t_bias[idx] = rb[j];
        t_lo[idx] = ok ? lf : 0.0f;
        t_hi[idx] = ok ? hf : __builtin_inff();
    }
    if constexpr (COMPAT) {
        s_remap[tid] = (unsigned char)r_remap;
        s_phi[tid] = r_phi;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    LN_STAMP(1);
    // (4) the first groups of the wave
    if constexpr (NG > 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) load_group(g, w[g]);
    } else {
        // double-buffered loop: the loop head must see the same operations behind the first buffer's loads from the prologue as
        // from the previous iteration (NC stores of the other buffer), or its wait would also wait for those stores' acknowledgement
        load_group(0, w[0]);
#pragma unroll
        for (int i = 0; i < NC; ++i)     // dropped by the range check; distinct offsets: identical stores would be merged into one
            __builtin_amdgcn_raw_buffer_store_b128(ln_v4u{0, 0, 0, 0}, ro, LN_OOR + (unsigned)(i * 16), 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);

    const float4* t_bias4 = reinterpret_cast<const float4*>(t_bias);
    const float4* t_lo4 = reinterpret_cast<const float4*>(t_lo);
    const float4* t_hi4 = reinterpret_cast<const float4*>(t_hi);

    // the arithmetic of one group, results in place of the inputs (no vector-memory operation except in the rare literal branch)
    auto compute = [&](int gi, ln_v4u (&wg)[NC]) {
        const int row = row_l0 + gi * RPG;
        const bool valid = row < a.rows;
        // ---- row sums
        int s1 = 0, s2 = 0, sq = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int v = (int)wg[i][d];
                if constexpr (COMPAT) {
                    sq = __builtin_amdgcn_sdot4(v, 0x01010101, sq, false);
                    const unsigned u = (unsigned)v ^ 0x80808080u;      // q + 128 per byte
                    v = (int)((unsigned)s_remap[u & 255] | ((unsigned)s_remap[(u >> 8) & 255] << 8) |
                              ((unsigned)s_remap[(u >> 16) & 255] << 16) | ((unsigned)s_remap[u >> 24] << 24));
                    wg[i][d] = (unsigned)v;
                }
                s1 = __builtin_amdgcn_sdot4(v, 0x01010101, s1, false);
                s2 = __builtin_amdgcn_sdot4(v, v, s2, false);
            }
        }
        s1 = ln_row_allreduce<LPR>(s1);
        s2 = ln_row_allreduce<LPR>(s2);
        // ---- statistics of this lane's row (ivit_modules.py:36-52)
        int mean_i;
        if constexpr (COMPAT) {
            sq = ln_row_allreduce<LPR>(sq);
            const float mf = (float)sq / (float)C;       // = fl(sum phi / C) unless the row is a tie (layernorm_i8_kernel)
            mean_i = (int)rintf(mf);
            // an exact .5 tie of sum q / C, with slack (a false positive only costs the exact sum below)
            const bool tie = fabsf(mf - floorf(mf) - 0.5f) <= 2.5f / (float)C && lq == 0 && valid;
            unsigned long long tm = __builtin_amdgcn_ballot_w64(tie);
            while (tm) {                       // wave-uniform, about 1 row in 300
                const int tl_ = __builtin_ctzll(tm);
                tm &= tm - 1;
                const int rr = tl_ / LPR;
                __builtin_amdgcn_wave_barrier();
                if (lr == rr) {                // the row's lanes re-read their (un-remapped) bytes into the wave's LDS row
                    const unsigned off = (unsigned)row * (unsigned)a.ldx + (unsigned)(lq * 16);
#pragma unroll
                    for (int i = 0; i < NC; ++i)
                        *reinterpret_cast<ln_v4u*>(&s_row[wave][(i * LPR + lq) * 16]) =
                            __builtin_amdgcn_raw_buffer_load_b128(rx, off + (unsigned)(i * LPR * 16), 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned char* rowb = s_row[wave];
                const float S = torch_rowsum_get([&](int i) { return s_phi[(int)(int8_t)rowb[i] + 128]; }, C, lane);
                const int fixed = (int)rintf(S / (float)C);     // ivit_modules.py:37
                mean_i = (lr == rr) ? fixed : mean_i;
            }
        } else {
            ln_mean(s1, C, mean_i);
        }
        const int var = s2 - 2 * mean_i * s1 + C * mean_i * mean_i;
        const float hfac = (abl & 2) ? 1.0f : ln_hfactor_small(var);
        const float mean128 = (float)(mean_i + 128);
        // ---- element chain, one 16-byte chunk at a time
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int q = i * LPR + lq;
            unsigned u = 0;
            ln_v4u res;
            float4 nb = t_bias4[q], nl = t_lo4[q], nh = t_hi4[q];      // the table reads run one dword ahead of the arithmetic
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float4 b4 = nb, l4 = nl, h4 = nh;
                if (d < 3) {
                    nb = t_bias4[(d + 1) * NQ + q];
                    nl = t_lo4[(d + 1) * NQ + q];
                    nh = t_hi4[(d + 1) * NQ + q];
                }
                const float bias[4] = {b4.x, b4.y, b4.z, b4.w}, lo[4] = {l4.x, l4.y, l4.z, l4.w}, hi[4] = {h4.x, h4.y, h4.z, h4.w};
                const unsigned wu = wg[i][d] ^ 0x80808080u;
                int ob[4];
                if (!(abl & 1)) {
                    // (v_pk_add / v_pk_mul / v_pk_fma cost 3.2 cycles per pair against 2 x 1.95 for the plain forms
                    // (profiles/r04_valu_price_list.txt), but the register pairs they need made the kernel spill at 128 VGPRs)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float xf = (float)((wu >> (8 * c)) & 0xffu);
                        const float dl = xf - mean128;                   // x - mean, exact
                        const float vv = floorf(dl * hfac);              // :52
                        const float y = vv + bias[c];                    // :61
                        const int tl = __float_as_int(__builtin_fmaf(y, lo[c], 12582912.0f));
                        const int th = __float_as_int(__builtin_fmaf(y, hi[c], 12582912.0f));
                        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(u) : "v"(tl), "v"(th), "v"(u));
                        ob[c] = clamp_i32(tl, 0x4B400000 - 128, 0x4B400000 + 127);   // low byte = int8 result
                    }
                } else {
                    ob[0] = ob[1] = ob[2] = ob[3] = (int)wu;
                }
                const unsigned w01 = __builtin_amdgcn_perm((unsigned)ob[1], (unsigned)ob[0], 0x0c0c0400u);
                const unsigned w23 = __builtin_amdgcn_perm((unsigned)ob[3], (unsigned)ob[2], 0x04000c0cu);
                res[d] = w01 | w23;
            }
            // literal evaluation of a chunk with an uncertified element (wave-uniform, ~1 % of the chunks)
            if (__builtin_amdgcn_ballot_w64(u != 0) != 0) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // buffer loads with 32-bit offsets: 64-bit lane addresses of three tables, hoisted out of the loop, spilled
                    const unsigned cho = (unsigned)(16 * q + 4 * d) * 4u;
                    const ln_v4u s4 = __builtin_amdgcn_raw_buffer_load_b128(r_sln, cho, 0, 0);
                    const ln_v4u m4 = __builtin_amdgcn_raw_buffer_load_b128(r_m, cho, 0, 0);
                    const ln_v4u e4 = __builtin_amdgcn_raw_buffer_load_b128(r_e, cho, 0, 0);
                    const float4 b4 = t_bias4[d * NQ + q];
                    const float sl[4] = {__uint_as_float(s4.x), __uint_as_float(s4.y), __uint_as_float(s4.z), __uint_as_float(s4.w)};
                    const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
                    const double Mq[4] = {dyadic_mult(m4.x, (int)e4.x), dyadic_mult(m4.y, (int)e4.y), dyadic_mult(m4.z, (int)e4.z),
                                          dyadic_mult(m4.w, (int)e4.w)};
                    int ob[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float dl = (float)(sx8((int)wg[i][d], c) - mean_i);
                        float v = floorf(dl * hfac);                       // :52
                        float y = v + bias[c];                             // :61
                        float x = y * sl[c];                               // :63
                        float qf = (float)((double)x * (1.0 / (double)sl[c]));   // quant_utils.py:220, see layernorm_i8_kernel
                        float z = rintf(qf);
                        double p = (double)z * Mq[c];                      // :229
                        double t = p + IVIT_MAGIC;                         // :230
                        ob[c] = clamp_i32((int)(unsigned)__double_as_longlong(t), -128, 127);
                    }
                    res[d] = (unsigned)pack4(ob[0], ob[1], ob[2], ob[3]);
                }
            }
            wg[i] = res;
        }
    };
    // stores (unconditional: a row that does not exist stores to LN_OOR, which the range check drops)
    auto store_group = [&](int gi, ln_v4u (&o)[NC]) {
        const int row = row_l0 + gi * RPG;
        const bool valid = gi < g_cnt && row < a.rows;
        if (a.out_blocks) {
            const unsigned rl = (unsigned)row & 15u, sw = (rl >> 2) & 3u;
            const unsigned base = (valid && !(abl & 4)) ? (unsigned)(row >> 4) * (unsigned)(C >> 6) * 1024u + (rl << 6) : LN_OOR;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const unsigned q = (unsigned)(i * LPR + lq);
                __builtin_amdgcn_raw_buffer_store_b128(o[i], ro, base + (q >> 2) * 1024u + (((q & 3u) ^ sw) << 4), 0, 0);
            }
        } else {
            const unsigned base = (valid && !(abl & 4)) ? (unsigned)row * (unsigned)a.ldo + (unsigned)(lq * 16) : LN_OOR;
#pragma unroll
            for (int i = 0; i < NC; ++i) __builtin_amdgcn_raw_buffer_store_b128(o[i], ro, base + (unsigned)(i * LPR * 16), 0, 0);
        }
    };

    // Rounds of NG groups: compute and store slot by slot, then reload all slots for the next round.  Every path through a
    // round issues the same vector-memory operations (a slot beyond the wave's last group skips the arithmetic only; its
    // stores and loads go to LN_OOR), and the loads in flight at the loop head are the same NG * NC whether the wave comes
    // from the prologue or from the previous round: the compiler's counted waits are exact (csrc/check_isa.py checks them).
    // Latency between rounds is covered by the other waves of the SIMD (four per SIMD).
    if constexpr (NG > 0) {
        for (int k = 0; k < g_cnt; k += NG) {
            // Issue priority by progress: the SIMD arbitrates by priority, then age, so at equal priority the four waves of a SIMD
            // finish one after another and the youngest ends alone, at a single wave's issue rate (4.9 instead of 1.95 / 3.2 cycles
            // per instruction).  A wave that is behind gets the higher priority: they finish together (20.2 vs 21.4 us, natural
            // scales 21.7 vs 23.6: profiles/r04i_*).  Lab bit 5: off.
            if (!(abl & 32)) {
                const int q4 = (4 * k) / g_cnt;
                if (q4 == 0) __builtin_amdgcn_s_setprio(3);
                else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
                else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int gi = k + g;
                if (gi < g_cnt) compute(gi, w[g]);
#if IVIT_LAB
                if (k == 0 && g < 3) LN_STAMP(2 + g);      // first round: slot g computed
#endif
                store_group(gi, w[g]);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) load_group(k + NG + g, w[g]);
        }
    } else {
        // NG = 0: one group in flight ahead of the one being computed (two register buffers).  All waves of the chip start
        // together and the kernel is bound by instruction issue: what matters is that EVERY wave gets its first rows quickly
        // (a round of two or three groups per wave up front put 25-38 MB into the memory queues at once, the last-served
        // waves got their first rows 9 us into the kernel: profiles/r04c_*, r04d_*), and that the next group is there when
        // the current one is done.
        for (int k = 0; k < g_cnt; k += 2) {
            load_group(k + 1, w[1]);
            if (!(abl & 32)) {  // issue priority by progress, as in the rounds loop above
                const int q4 = (4 * k) / g_cnt;      // quarter of the wave's groups it is in
                if (q4 == 0) __builtin_amdgcn_s_setprio(3);
                else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
                else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            compute(k, w[0]);
#if IVIT_LAB
            if (k == 0) LN_STAMP(2);
#endif
            store_group(k, w[0]);
            load_group(k + 2, w[0]);
            if (k + 1 < g_cnt) compute(k + 1, w[1]);
#if IVIT_LAB
            if (k == 0) LN_STAMP(3);
#endif
            store_group(k + 1, w[1]);
        }
    }
#if IVIT_LAB
    if (a.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[6] = __builtin_amdgcn_s_memrealtime();
        stamp[5] = __builtin_amdgcn_s_memtime() - clk0;      // shader cycles of the wave's lifetime: clock = stamp[5] / (stamp[6] - stamp[0]) x 100 MHz
        stamp[7] = (unsigned long long)g_cnt;
        if (lane == 0)
            for (int i = 0; i < 8; ++i) a.stamps[(size_t)wave_id * 8 + i] = stamp[i];
    }
#endif
#undef LN_STAMP
}

// shapes the streaming kernel takes: C = LPR * 16 * NC exactly, 16-byte aligned rows on both sides
static inline bool ln_stream_takes(const LnArgs& a)
{
    const int C = a.C;
    if (!(C == 192 || C == 384 || C == 768 || C == 512 || C == 1024)) return false;
    // 32-bit buffer offsets: both matrices below 2 GiB
    return a.ldx % 16 == 0 && a.ldo % 16 == 0 && ((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.out % 16 == 0) && a.outer == 0 &&
           (int64_t)a.rows * a.ldx < 2147483648ll && ((int64_t)a.rows + 15) * (a.out_blocks ? (int64_t)a.C : a.ldo) < 2147483648ll;
}

static inline bool ln_stream_pays(const LnArgs& a) { return (int64_t)a.rows * a.C >= 12000000; }

template <bool COMPAT>
static int launch_ln_stream(const LnArgs& a, hipStream_t st, const char* who, int cfg)
{
    // rows per group and a grid of at most 2 workgroups per CU with at least ~3 groups per wave
    const int lpr = a.C == 192 ? 4 : a.C == 384 ? 8 : 16;
    const int rpg = 64 / lpr;
    const int64_t n_groups = ((int64_t)a.rows + rpg - 1) / rpg;
    int grid = (int)((n_groups + 3 * WPB - 1) / (3 * WPB));
    // lab: cfg bits 0-3 kernel variant (0 = default), bits 4-7 workgroups per CU (0 = 4)
    const int wg_per_cu = (IVIT_LAB && (cfg >> 4)) ? (cfg >> 4) & 15 : 4;
    const int max_grid = 256 * wg_per_cu;
    if (grid > max_grid) grid = max_grid;
    if (grid < 1) grid = 1;
    const size_t lds = wg_per_cu == 1 ? (size_t)(60 * 1024) : (size_t)(160 * 1024 / (wg_per_cu + 1) + 1024);   // one more does not fit a CU
    cfg &= 15;
#define IVIT_LN_STREAM(LPRv, NCv, NGv, OCCv) \
    hipLaunchKernelGGL((layernorm_i8_stream_kernel<LPRv, NCv, NGv, COMPAT, OCCv>), dim3(grid), dim3(NT), lds, st, a)
    if (a.C == 768) {
#if IVIT_LAB
        if (cfg == 1) IVIT_LN_STREAM(16, 3, 4, 2);         // run with bits 4-7 = 2: two workgroups per CU, four groups per round
        else if (cfg == 3) IVIT_LN_STREAM(16, 3, 3, 4);
        else if (cfg == 4) IVIT_LN_STREAM(16, 3, 2, 4);
        else if (cfg == 5) IVIT_LN_STREAM(16, 3, 0, 4);
        else if (cfg == 6 && !COMPAT) IVIT_LN_STREAM(16, 3, 1, 6);     // run with bits 4-7 = 6: six workgroups per CU (<= 85 VGPRs)
        else if (cfg == 7 && !COMPAT) IVIT_LN_STREAM(16, 3, 1, 5);     // ... = 5
        else
#endif
        IVIT_LN_STREAM(16, 3, 1, 4);
    } else if (a.C == 384) IVIT_LN_STREAM(8, 3, 1, 4);
    else if (a.C == 192) IVIT_LN_STREAM(4, 3, 1, 4);
    else if (a.C == 512) IVIT_LN_STREAM(16, 2, 1, 4);
    else IVIT_LN_STREAM(16, 4, 1, 4);
#undef IVIT_LN_STREAM
    IVIT_CHECK_LAUNCH(who);
}
