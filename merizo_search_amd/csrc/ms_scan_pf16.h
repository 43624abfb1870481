// The prefilter's scan over the FP16 IMAGE of the database (ms_pf_build_image with MS_PF_F16X2 / MS_PF_F16X1): round 5.
//
// Round 4's split-bf16 image (ms_scan_pf.h) spends THREE matrix instructions per 16 dimensions and 512 B per row on an error bound of
// 2.5e-4 |row||q|.  An fp16 operand has an 11-bit significand: rounding the ROW to fp16 (to nearest even) costs 2^-11 |row_i| per
// component, and with the QUERY split hi / lo (both fp16: q = qh + ql to 2^-22) the scan needs TWO v_mfma_f32_32x32x16_f16 per 16
// dimensions (rowh.qh + rowh.ql) and 256 B per row for |a - s| <= 5.5e-4 |row||q| (MS_PF_ERR_F16X2); with the query's hi part alone ONE
// instruction per 16 dimensions for <= 1.05e-3 |row||q| (MS_PF_ERR_F16X1).  Half the image memory, half the HBM bytes, 2/3 or 1/3 of
// the matrix work; the exact re-scoring, the proof of completeness and the exact pass behind it are unchanged (ms_search.hip: pf_run),
// only E is larger -- on i.i.d. data the gap between a query's k-th and 2k-th best scores is ~1e-2, so nothing more is flagged.
//   Error budget, per unit of |row||q| (B = the database's row-norm bound; a = what the matrix pipe accumulates, s = the fp32 chain of
//   the exact scan):  row rounding 2^-11 = 4.883e-4;  [F16X1: query rounding 2^-11 more, cross term 2^-22]  query residual 2^-22 =
//   2.4e-7;  components that fall below the fp16 normal range AFTER scaling (row_i < 2^-28 B, q_i < 2^-27 max|q|), flushed or not:
//   2 x 8.4e-8;  fp32 accumulation of 256 (128) products inside the matrix pipe, truncating at worst: 256 x 2^-23 = 3.05e-5;  the
//   exact chain's own 128 roundings: 7.6e-6;  [round 6, raw queries normalised in the scan's set-up: q * (1 / max(|q|, eps)) instead of
//   F.normalize's q / max(|q|, eps), the sum of squares in another order: <= 4 ulp of fp32 = 4.8e-7].  Sum 5.27e-4 (F16X2), 1.00e-3 (F16X1);
//   MS_PF_ERR_* round up.  tools/stress_prefilter.py
//   measures max |a - s| / (|row||q|) over random shapes against these.
//   Range: the image holds row * 2^sr, sr = 14 - floor(log2 B), so every component is below 2^15 (clamped to +-65504 should B have
//   been wrong: an fp16 inf would turn a row's scores into NaN, which no filter passes); each query goes in as q * 2^sq with its largest
//   component in [2^13, 2^14).  Products stay below 2^29, sums below 2^36; the accumulators hold score * 2^(sr + sq) and are compared
//   against thresholds in that domain; a candidate's score is scaled back (exactly) in the rare path.  B outside [2^-40, 2^40]: no
//   fp16 image (ms_pf_build_image declines); a query whose norm is outside that range is flagged for the exact pass by the re-scoring.
//
// Image: tile T = rows 64 T .. 64 T + 63 (zero rows past the end) = 16 KiB at byte 16384 T; inside it fragment f = 2 b + half (k block
// b = 0..7, half = rows 0-31 / 32-63) = 1 KiB at 1024 f, lane (r, h) = 16 bytes at 16 (32 h + r): the eight fp16 of row 32 half + r at
// dimensions 64 h + 8 b + j.  256 B per row; a 256-byte trailer behind the last tile carries {magic, sr, n}.  So a tile is again
// sixteen linear 1 KiB LDS-DMA pieces and sixteen ds_read_b128 fragments: the ring, the arrival counters, the piece schedule and the
// shared bound of ms_scan_pf2_kernel are used UNCHANGED -- a stage now covers 64 rows, with two accumulator chains (one per half
// tile), which also halves everything a stage pays per tile that is not a matrix instruction (the bf16 stage is issue-bound:
// DESIGN.md 5.5).  Per stage and wave: 32 (F16X2) or 16 (F16X1) matrix instructions for 64 rows x 32 queries.
#pragma once
#include "ms_scan_pf.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr uint32_t MS_PF16_MAGIC = 0x3631464du;       // "MF16"
#define MS_PF_ERR_F16X2 5.5e-4f
#define MS_PF_ERR_F16X1 1.05e-3f
#ifndef MS_PF16_SHADOW
#define MS_PF16_SHADOW 0
#endif
// Ablation builds (WRONG results; tools/pf_scan_only.py): what does each part of the steady stage cost?
//   MS_PF16_ABL_NOMFMA  no matrix instructions (the accumulators are kept alive through an empty asm)
//   MS_PF16_ABL_NOFILTER  no lane maxima / threshold compare (the rare path is never taken)
//   MS_PF16_ABL_NODMA  no LDS-DMA pieces after the prologue (stale tiles; publication and waits unchanged)
#ifdef MS_PF16_HALF_LDS          // (diagnostic build, WRONG results: only half of a tile's fragments are read from LDS -- what do the LDS reads cost?)
#define MS_PF16_SECOND_READ(B)
#else
#define MS_PF16_SECOND_READ(B) fr[2 * (B) + 1] = src[64 * (2 * (B) + 1)];
#endif
// Round 6 (MS_PF16_DYNPRIO, default on): the wave that finds its next tile complete on arrival -- the one the workgroup is waiting for, e.g.
// after a visit of the rare path -- takes the higher issue priority for its next chain, a wave that has to wait the lower one: -2 to -2.5 % on
// long streams in two interleaved same-box passes (16M x 1024: 3.48 -> 3.40 ms; 4M x 256: 0.252 -> 0.246 ms), C2 unchanged
// (profiles/r06_pf16_dynprio_ab.log)
#ifndef MS_PF16_DYNPRIO
#define MS_PF16_DYNPRIO 1
#endif
#ifndef MS_PF16_ILV
#define MS_PF16_ILV 0
#endif
#if MS_PF16_DYNPRIO
#define MS_PF16_PRIO_BEHIND "s_setprio 2\n\t"
#define MS_PF16_PRIO_AHEAD "s_setprio 0\n\t"
#else
#define MS_PF16_PRIO_BEHIND
#define MS_PF16_PRIO_AHEAD
#endif
#ifndef MS_PF16_HIST_PERIOD
#define MS_PF16_HIST_PERIOD MS_HIST_PERIOD     // tiles between two looks at the shared bound in this kernel (8 needs MS_PF2_HIST_AREAS = 8)
#endif
#ifndef MS_PF16_LEAD_LOAD
#define MS_PF16_LEAD_LOAD 0        // 1: eight-wave workgroups, only waves 0-3 issue the LDS-DMA pieces -- measured: no change on five shapes
                                   // (profiles/r06_pf16_lead_load_ab.log); 0 (default): every wave two pieces, as in rounds 4-5
#endif
#ifndef MS_PF16_VISIT2
#define MS_PF16_VISIT2 1           // the rare path as straight-line predicated code (0: the ballot-and-select-tree form of rounds 4-5)
#endif
#ifndef MS_PF16_GROUPED
#define MS_PF16_GROUPED 1          // the rare path looks for candidates group of four registers by group (0: sixteen ballots, rounds 4-5)
#endif

// The approximate score of (row, query): one accumulator chain per half tile, k blocks in order, per block rowh.qh [, rowh.ql].
// The sample pass and the full pass run exactly this sequence (the sample's bound must hold bit for bit).
// MASK: MS_MODE_COSINE_UNIT with a length mask (p.lengths != NULL).  NQP: 2 = the query split hi / lo (MS_PF_F16X2), 1 = hi only.
template <int KL, int NW, bool SAMPLE, bool MASK, int NQP>
__global__ __launch_bounds__(64 * NW, NW / 4) void ms_scan_pf16_kernel(const ScanParams p) {
    static_assert(NQP == 1 || NQP == 2, "query parts");
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    // Round 6 experiment (MS_PF16_LEAD_LOAD=1, not the default: no gain): with eight waves only the FOUR OLDER ones (w < 4: they win the matrix pipe, run ahead and wait ~500 cycles per
    // tile for the others) issue the LDS-DMA pieces, four each; the younger four -- the critical path -- issue none and only vouch for
    // their progress at the arrival counters.  The requests also leave earlier (the leaders are up to two tiles ahead).
    constexpr int LW = (NW == 8 && MS_PF16_LEAD_LOAD) ? 4 : NW;      // waves that load
    constexpr int PPW = 16 / LW;                       // LDS-DMA pieces of a tile per loading wave
#ifdef MS_STAMP
    const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();
    unsigned long long tl_setup = 0, tl_first = 0, tl_loop = 0;
#endif
    if (ms_gate_closed(p.gate, p.gate_epoch)) return;          // (uniform: a scalar load)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char_t;
    typedef volatile __attribute__((address_space(3))) uint32_t lds_flag_t;
    typedef __attribute__((address_space(3))) ms_u32x2 lds_cand_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    lds_flag_t *arrived = (lds_flag_t *)((lds_char_t *)smem + PF2_OFF_CNT);     // [16]
    const float *auxring = reinterpret_cast<const float *>(smem + PF2_OFF_AUX);

    const int bid = blockIdx.x;
    const int per_super = 8 * p.n_qgroups;
    const int super = bid / per_super, within = bid % per_super;
    const int stream = super * 8 + (within & 7);      // one stream per workgroup; workgroups 8 apart share an XCD (and its L2)
    const int qg = within >> 3;
    if (stream >= p.n_streams) return;
    const int64_t row_begin = (int64_t)stream * p.rows_per_stream;
    const int64_t row_end = (row_begin + p.rows_per_stream < p.n) ? row_begin + p.rows_per_stream : p.n;
    const int nfull = (int)((row_end - row_begin) >> 6);        // 64-row tiles
    const int rem = (int)((row_end - row_begin) & 63);
    const int ntl = SAMPLE ? (nfull < p.max_tiles ? nfull : p.max_tiles) : nfull + (rem > 0 ? 1 : 0);
    constexpr bool mask_on = MASK;                      // cosine on unit rows

    // query tile of wave w: qg * NW + w; a wave without a real one only loads its pieces of the tiles
    const int qtile = qg * NW + wave;
    const bool has_q = qtile < p.n_qtiles;
    if (tid < PF2_ARR) arrived[tid] = 0u;
    __syncthreads();
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char_t *)smem);

    // ---- this wave's pieces of tile t -> slot t % R (wave 0, cosine mode: one more, the rows' lengths)
    const uint32_t voff = (uint32_t)(16 * lane);
    const uint64_t img0 = (uint64_t)(uintptr_t)p.pf_image + (uint64_t)(row_begin >> 6) * 16384u + (uint32_t)(1024 * PPW) * (uint32_t)wave;
    // (cosine mode: EVERY wave issues one more piece per tile so that the counted waits are the same for all of them; only wave 0's
    //  -- the rows' lengths -- is read)
#ifdef MS_PF16_ABL_NODMA
    bool abl_prologue_done = false;
#endif
    uint64_t it_sb = 0;            // base address and LDS destination of the tile being issued (uniform)
    uint32_t it_dst = 0;
    const bool loads = (LW == NW) ? true : (wave < LW);      // (uniform; a constant when every wave loads: no branch around the pieces)
    auto issue_prep = [&](int t) __attribute__((always_inline)) {
        if (!loads) return;
        const uint64_t b = img0 + (uint64_t)t * 16384u;
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        it_sb = ((uint64_t)b_hi << 32) | (uint64_t)b_lo;
        it_dst = (uint32_t)__builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)(t % PF2_R) * 16384u + (uint32_t)(1024 * PPW) * (uint32_t)wave);
    };
    auto issue_piece = [&](auto i_c) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value;
        if constexpr (I < PPW) {
            if (!loads) return;
            // (uniform values that live across branches: say so again, or the "s" operands of the asm may be handed vector registers)
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane(it_dst);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)it_sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(it_sb >> 32));
#ifdef MS_PF16_ABL_NODMA
            if (abl_prologue_done) { asm volatile("" :: "s"(d), "s"(lo), "s"(hi)); return; }
#endif
#ifdef MS_PF2_NT
            ms_glds_s16_nt<1024 * I>(d + 1024 * I, voff, ((uint64_t)hi << 32) | (uint64_t)lo);
#else
            ms_glds_s16<1024 * I>(d + 1024 * I, voff, ((uint64_t)hi << 32) | (uint64_t)lo);
#endif
        }
    };
    auto issue_aux = [&](int t) __attribute__((always_inline)) {
        if constexpr (MASK) {
            if (!loads) return;
            int64_t row = row_begin + (int64_t)t * 64 + lane;        // (64 rows per tile: one length per lane)
            if (row >= p.n) row = p.n - 1;
            const uint32_t dst = wave == 0 ? ring_lds + PF2_OFF_AUX + (uint32_t)(t % PF2_AUXR) * 256u
                                           : ring_lds + PF2_OFF_DUMMY + (uint32_t)(wave - 1) * 256u;
            ms_glds_v4((uint32_t)__builtin_amdgcn_readfirstlane(dst), p.lengths + row);
        }
    };
    auto issue_tile = [&](int t) __attribute__((always_inline)) {
        issue_prep(t);
        issue_piece(std::integral_constant<int, 0>{}); issue_piece(std::integral_constant<int, 1>{});
        issue_piece(std::integral_constant<int, 2>{}); issue_piece(std::integral_constant<int, 3>{});
        issue_aux(t);
    };
    // own pieces of every tile but the youngest N issued have landed (a wave that loads nothing has nothing to wait for)
    auto wait_own = [&](auto n_c) __attribute__((always_inline)) {
        constexpr int N = decltype(n_c)::value;
        if (loads) ms_pf2_vmcnt<(PPW + (MASK ? 1 : 0)) * N>();
    };
    // ... the same in front of the shared bound's staging area (its two pieces are this wave's own whether it loads tiles or not)
    auto wait_own_hist = [&]() __attribute__((always_inline)) {
        if (loads) ms_pf2_vmcnt<(PPW + (MASK ? 1 : 0)) * 2>(); else ms_pf2_vmcnt<0>();
    };
    // publication: this wave's pieces of tile t have landed -> one more arrival at the tile's counter
    const uint32_t arr_lds = ring_lds + PF2_OFF_CNT;
    auto publish = [&](int t) __attribute__((always_inline)) {
#ifdef MS_PF2_PUBLISH_C
        if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)(arrived + (t & (PF2_ARR - 1))), 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        // (one lane adds, under an EXEC mask set by scalar moves -- the code is wave-uniform here, EXEC is all ones: five
        //  instructions; hipcc's `if (lane == 0) atomic add` is fifteen, with two branches, in every stage)
        const uint32_t a_ = arr_lds + 4u * (uint32_t)(t & (PF2_ARR - 1));
        uint32_t pub_a, pub_one;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, 1\n\ts_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1"
                     : "=&v"(pub_a), "=&v"(pub_one) : "s"(a_) : "memory");
#endif
    };
    // every wave's pieces of tile t have landed: its counter has been raised NW times per use of it
    uint32_t seen = 0;                                  // the counter of the tile the next stage needs, as last read
    auto read_arrived = [&](int t) __attribute__((always_inline)) { seen = arrived[t & (PF2_ARR - 1)]; };
    auto wait_arrived = [&](int t) __attribute__((always_inline)) {
        const uint32_t need = (uint32_t)NW * (uint32_t)(t / PF2_ARR + 1);
#ifdef MS_PF2_WAIT_C
        uint32_t spins = 0;
#pragma unroll 1
        for (; (uint32_t)__builtin_amdgcn_readfirstlane(seen) < need && spins < (1u << 24); ++spins) {
            __builtin_amdgcn_s_sleep(1);
            read_arrived(t);
        }
        if (__builtin_expect(spins >= (1u << 24), 0)) __builtin_trap();      // never a silent hang
#else
        // ONE asm statement: the snapshot is good -> four instructions and a short forward branch (hipcc's loop around the same
        // test is nineteen instructions with a taken branch even when there is nothing to wait for).  Bounded: never a silent hang.
        uint32_t sv_, spins_, av_;
        static_assert(PF2_ARR == 16, "the mask below");
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %0, %2\n\t"
                     "s_cmp_ge_u32 %0, %5\n\t"
                     MS_PF16_PRIO_BEHIND                   // (nothing to wait for: this wave is the one the others wait for)
                     "s_cbranch_scc1 2f\n\t"
                     MS_PF16_PRIO_AHEAD                    // (it has to wait: it is ahead)
                     "s_and_b32 %0, %4, 15\n\t"           // (the counter's address: only needed on this path)
                     "s_lshl_b32 %0, %0, 2\n\t"
                     "s_add_u32 %0, %0, %6\n\t"
                     "v_mov_b32 %3, %0\n\t"
                     "s_mov_b32 %1, 0\n\t"
                     "1:\n\t"
                     "s_sleep 1\n\t"
                     "ds_read_b32 %2, %3\n\t"
                     "s_add_u32 %1, %1, 1\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %0, %2\n\t"
                     "s_cmp_ge_u32 %0, %5\n\t"
                     "s_cbranch_scc1 2f\n\t"
                     "s_cmp_lt_u32 %1, 0x1000000\n\t"
                     "s_cbranch_scc1 1b\n\t"
                     "s_trap 2\n\t"                        // never a silent hang
                     "2:"
                     : "=&s"(sv_), "=&s"(spins_), "+v"(seen), "=&v"(av_) : "s"(t), "s"(need), "s"(arr_lds) : "memory", "scc");
#endif
        asm volatile("" ::: "memory");          // (the tile's fragment reads stay behind the wait)
    };
    (void)arr_lds;
    // ---- set-up loads: ALL requested here, in front of the prologue's LDS-DMA pieces, and consumed behind them, so that they share ONE
    //      round trip with the first tiles (measured neutral against requesting them behind the prologue: 54.9 / 110.9 us against 54.4 /
    //      111.7 at 140k / 1M rows -- the ~7.5 us between a wave's entry and its first stage at C2 are not these loads)
    const int qidx = qtile * 32 + r;
    const bool q_valid = qidx < p.nq;
    const bool hist_on = !SAMPLE && p.hist != nullptr && p.lb_s != nullptr;      // (uniform)
    f32x4 qv[16];
    float pre_lb = -INFINITY, pre_stp = 0.0f, pre_qlen = 0.0f;
    uint32_t pre_magic = MS_PF16_MAGIC, pre_n = (uint32_t)p.n;
    int pre_sr = 0;
    if (has_q) {
        const uint32_t *trailer = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p.pf_image) + (size_t)((p.n + 63) >> 6) * 16384u);
        pre_magic = trailer[0];
        pre_sr = (int)trailer[1];
        pre_n = trailer[2];
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.qn + (size_t)(q_valid ? qidx : 0) * MS_DIM + 64 * h);
#pragma unroll
        for (int i = 0; i < 16; ++i) qv[i] = src[i];
        if (!SAMPLE && p.lb_s != nullptr) pre_lb = p.lb_s[qidx];
        if (hist_on && q_valid) pre_stp = p.hstep[qidx];
        if (MASK && p.qlen != nullptr && q_valid) pre_qlen = p.qlen[qidx];
    }
    asm volatile("" ::: "memory");          // (the loads stay in front of the DMA pieces below)
    // ---- prologue: the first D tiles are requested before anything else (HBM latency overlaps the query set-up)
#pragma unroll
    for (int t = 0; t < PF2_D; ++t)
        if (t < ntl) issue_tile(t);

#ifdef MS_PF16_ABL_NODMA
    abl_prologue_done = true;
#endif
    if (!has_q) {
        // loading-only wave (the workgroup's last query tiles are padding): issue, publish, keep pace with the readers
        // (it waits for the same arrivals as a wave that computes: that is what keeps it from overwriting a slot in use)
        if (ntl > 0) {
            if (ntl >= PF2_D) wait_own(std::integral_constant<int, PF2_D - PF2_W>{}); else ms_pf2_vmcnt<0>();
#pragma unroll
            for (int t = 0; t < PF2_W; ++t)
                if (t < ntl) publish(t);
        }
        for (int t = 0; t < ntl; ++t) {
            if (t + 1 < ntl) { read_arrived(t + 1); wait_arrived(t + 1); }
            if (t + PF2_D < ntl) {
                issue_tile(t + PF2_D);
                wait_own(std::integral_constant<int, PF2_D - PF2_W>{});   // own pieces of tiles <= t + W have landed
            } else {
                ms_pf2_vmcnt<0>();
            }
            if (t + PF2_W < ntl) publish(t + PF2_W);
        }
        return;
    }

    // ---- compute waves: queries, lists, bounds
    ScanState<SAMPLE ? 1 : KL> st;
    ScanHist hg;
    f16x8 qh[8], ql[NQP == 2 ? 8 : 1];
#pragma unroll
    for (int j = 0; j < (SAMPLE ? 1 : KL); ++j) { st.ls[j] = -INFINITY; st.li[j] = MS_IDX_NONE; }
    st.floor = -INFINITY;
    st.tau = -INFINITY;
    if (!SAMPLE && p.lb_s != nullptr) {
        const float lb = pre_lb;
        st.floor = (lb == -INFINITY) ? -INFINITY : nextafterf(lb, -INFINITY);
        st.tau = st.floor;
    }
    if (!q_valid) { st.floor = INFINITY; st.tau = INFINITY; }   // padding queries never pass the filter
    if (!SAMPLE && (p.debug_flags & 1)) { st.floor = INFINITY; st.tau = INFINITY; }      // (diagnostics: nothing ever passes)
    hg.counters = nullptr; hg.base = 0.0f; hg.step = 0.0f; hg.inv_step = 0.0f;
    if (hist_on && q_valid) {
        const float stp = pre_stp, lb = pre_lb;
        if (stp > 0.0f && lb > -INFINITY) { hg.counters = p.hist + (size_t)qidx * 16; hg.base = lb; hg.step = stp; hg.inv_step = 1.0f / stp; }
    }
    // Scales.  The image holds row * 2^sr as fp16 (sr from the database's row-norm bound: components below 2^15; the image's trailer
    // carries it), the query goes in as q * 2^sq with its largest component in [2^13, 2^14): nothing that matters is subnormal on
    // either side, no product overflows, and the accumulators hold score * 2^(sr + sq) -- thresholds are compared in that domain
    // (tau_s), scores leave it (exactly: a power of two) only in the rare path.
    float up = 1.0f, down = 1.0f;
    {
        const uint32_t magic = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre_magic);
        const int sr = __builtin_amdgcn_readfirstlane(pre_sr);
        const uint32_t img_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre_n);
        // not an fp16 image, or the image of a database with another row count (n < 2^31): never a silent wrong answer
        if (magic != MS_PF16_MAGIC || img_n != (uint32_t)p.n) __builtin_trap();
        f32x4 v[16];
        float m = 0.0f;
        float rinv = 1.0f;
        if (p.qraw_eps > 0.0f) {            // (uniform) raw queries: q / max(|q|, eps) to a few ulp -- this lane's 64 squares in order, then the partner's
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ss += qv[i].x * qv[i].x; ss += qv[i].y * qv[i].y; ss += qv[i].z * qv[i].z; ss += qv[i].w * qv[i].w; }
            ss = ss + ms_xor32_f(ss, h);
            rinv = 1.0f / fmaxf(sqrtf(ss), p.qraw_eps);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = q_valid ? qv[i] * rinv : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        }
        m = fmaxf(m, ms_xor32_f(m, h));
        int sq = 0;
        if (m > 0.0f && m < INFINITY) sq = 13 - ((int)((__float_as_uint(m) >> 23) & 0xFFu) - 127);
        sq = sq > 60 ? 60 : (sq < -60 ? -60 : sq);
        const float qs = __uint_as_float((uint32_t)(127 + sq) << 23);
        up = __uint_as_float((uint32_t)(127 + sr + sq) << 23);         // (|sr| <= 40, |sq| <= 60)
        down = __uint_as_float((uint32_t)(127 - sr - sq) << 23);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const f32x4 a0 = v[2 * b] * qs, a1 = v[2 * b + 1] * qs;
            const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = fminf(fmaxf(x[j], -65504.0f), 65504.0f);
#ifdef MS_PF16_ABL_BF16
                const _Float16 hi16 = __builtin_bit_cast(_Float16, (__bf16)c);
#elif defined(MS_PF16_ABL_F16R8)
                const _Float16 hi16 = (_Float16)(float)(__bf16)c;
#else
                const _Float16 hi16 = (_Float16)c;                     // round to nearest even
#endif
                qh[b][j] = hi16;
                if constexpr (NQP == 2) ql[b][j] = (_Float16)(c - (float)hi16);
            }
        }
    }
    float tau_s = st.tau * up;           // st.tau in the accumulators' domain (-inf / +inf stay what they are)
    const float my_qlen = (mask_on && p.qlen != nullptr && q_valid) ? pre_qlen : 0.0f;
    const float qlen_eff = mask_on ? my_qlen : INFINITY;
    const float mincov_eff = mask_on ? p.mincov : 0.0f;
    float smax = -INFINITY;
    // cosine mode: scores of tile t (16 per lane: row 8 g + 4 h + j in register 4 g + j) times the length mask of their rows
    auto apply_mask = [&](f32x16 &acc, int t, int half) __attribute__((always_inline)) {
        const f32x4 *ax = reinterpret_cast<const f32x4 *>(auxring + (t & (PF2_AUXR - 1)) * 64 + 32 * half + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 len4 = ax[2 * g];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mk = (qlen_eff >= len4[j] * mincov_eff) ? 1.0f : 0.0f;           // dbsearch.py:76
                acc[4 * g + j] = acc[4 * g + j] * mk;                                           // dbsearch.py:78
            }
        }
    };
    bool neg_tau = !SAMPLE && mask_on && (__ballot(st.tau < 0.0f) != 0);
    // ---- pacing (several query groups per row stream: C4's 4096 queries are 16 workgroups on every stream, all on one XCD).  Left
    //      alone they drift apart -- one visits the rare path more often, another loses the matrix pipe less -- and each then fetches
    //      the stream's tiles from HBM for itself: 9.4 x the image per launch at the C4 shard (profiles/r05_pf_c4_pmc_free.json).
    //      Wave 0 of every workgroup publishes its tile number every 16 tiles (one word per workgroup, the 16 words of a stream in
    //      ONE 64-byte line; tagged with the launch's 8-bit epoch: no clearing between launches) and reads the line back with a SCALAR load
    //      (glc: from the L2 all of them share; its own counter, so the vector-memory queue with the tiles in flight is not
    //      drained): a workgroup more than PACE_LEAD tiles ahead of the slowest sleeps until it is not (bounded: pacing is an
    //      optimisation, never a condition for progress).  The other waves of the workgroup follow through the arrival counters.
    const bool pace_on = !SAMPLE && p.prog != nullptr;          // (uniform)
    constexpr int PACE_LEAD = 48;
    int pace_left = 16384;          // sleeps of ~2 us this workgroup may spend waiting in all (32 ms): should the workgroups of a stream NOT be
                                    // co-resident (fewer CUs than the plan assumed), the leaders give up pacing instead of waiting for a
                                    // workgroup that has not started
    auto pace = [&](int t) __attribute__((always_inline)) {
        if (wave != 0 || (t & 15) != 0) return;
        const uint64_t line = (uint64_t)(uintptr_t)(p.prog + (size_t)stream * 16);
        const uint32_t l_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)line), l_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(line >> 32));
        const uint64_t sline = ((uint64_t)l_hi << 32) | (uint64_t)l_lo;
        const uint32_t tag = p.prog_epoch << 24;                   // (8-bit epoch, 24-bit tile number: a stream has < 2^24 tiles)
        if (lane == 0) __hip_atomic_store(p.prog + (size_t)stream * 16 + qg, tag | (uint32_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < PACE_LEAD) return;
        typedef uint32_t u32x16_ __attribute__((ext_vector_type(16)));
        for (; pace_left > 0; --pace_left) {
            u32x16_ w;
            asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(sline) : "memory");
            uint32_t mn = 0xFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t v = ((w[i] >> 24) == p.prog_epoch) ? (w[i] & 0xFFFFFFu) : 0u;        // another launch's word: not started yet
                if (i < p.n_qgroups) mn = v < mn ? v : mn;
            }
            if ((uint32_t)t <= mn + (uint32_t)PACE_LEAD) break;
            __builtin_amdgcn_s_sleep(64);
        }
    };
    auto retau = [&]() __attribute__((always_inline)) { tau_s = st.tau * up; };

    // rare path: the candidates of tile t (scores sc_v) are counted and buffered; the lists take them later
    uint32_t ccnt = 0;
    auto cand_slot = [&](uint32_t c) __attribute__((always_inline)) -> lds_cand_t * {
        return (lds_cand_t *)((lds_char_t *)smem + PF2_OFF_CAND + (wave * PF2_CAND) * 512) + c * 64 + lane;
    };
    auto flush = [&]() __attribute__((always_inline)) {
        for (uint32_t c = 0; __ballot(ccnt > c) != 0; ++c) {
            const ms_u32x2 e = *cand_slot(c < PF2_CAND ? c : 0);
            const float v = (ccnt > c) ? __uint_as_float(e.x) : -INFINITY;
            ms_lane_insert<SAMPLE ? 1 : KL>(st, v, e.y, 0, h);
            ms_lane_insert<SAMPLE ? 1 : KL>(st, v, e.y, 1, h);
        }
        ccnt = 0;
        retau();
    };
    // Round 6: the rare path as straight-line predicated code.  The form below it (rounds 4-5) finds the candidate registers with a chain of
    // ballots, then loops over them with a uniform index and a 15-select tree: ~130 instructions of which most are scalar-after-vector
    // dependencies (v_cmp -> s_cmp -> s_cselect -> s_or ...), 1,400 cycles per visit -- and a visit of ANY of the eight waves delays the whole
    // workgroup (no wave has spare speed to catch up: the diagnostic build without visits runs 19-27 % faster, profiles/r06_pf16_diag_*).
    // Here: group maxima (8 instructions), one test per group of four registers, and per register of a hit group `v_cmp; s_and_saveexec;
    // s_cbranch_execz` around a body that runs for the passing lanes only (count in the histogram, append, mark the register taken).  A lane
    // whose buffer is full does not append: it raises `ovf`, the buffers are flushed into the lists and the tile is visited again -- the
    // registers already taken are -inf by then (the accumulators are dead after the visit: the next chain zeroes them).
    // Measured (profiles/r06_pf16_visit2_ab.log): k = 10 shapes unchanged (the visits that actually run cost 8 % of the C2 launch and 2 % of
    // a long stream: MS_PF_DEBUG=1), k = 32 (16-entry lists) 6 % faster.  The cosine + mask instantiations with the query split hi / lo or with
    // 16-entry lists spill 330-1,250 bytes per lane around this form and keep the older one.
    constexpr bool VISIT2 = MS_PF16_VISIT2 && !(MASK && (NQP == 2 || KL >= 16));
    auto visit_v2 = [&](f32x16 &sc_v, int t, int half, bool check_rows) __attribute__((always_inline)) {
        if (mask_on) apply_mask(sc_v, t, half);
        const uint32_t sub_row0 = (uint32_t)(row_begin + (int64_t)t * 64) + (uint32_t)(32 * half + 4 * h);
#pragma unroll 1
        for (;;) {
            bool ovf = false;
            float g0, g1, g2, g3;
            asm("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %8, %9, %10\n\tv_max3_f32 %2, %12, %13, %14\n\tv_max3_f32 %3, %16, %17, %18\n\t"
                "v_max_f32 %0, %0, %7\n\tv_max_f32 %1, %1, %11\n\tv_max_f32 %2, %2, %15\n\tv_max_f32 %3, %3, %19"
                : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
                : "v"(sc_v[0]), "v"(sc_v[1]), "v"(sc_v[2]), "v"(sc_v[3]), "v"(sc_v[4]), "v"(sc_v[5]), "v"(sc_v[6]), "v"(sc_v[7]), "v"(sc_v[8]),
                  "v"(sc_v[9]), "v"(sc_v[10]), "v"(sc_v[11]), "v"(sc_v[12]), "v"(sc_v[13]), "v"(sc_v[14]), "v"(sc_v[15]));
#define MS_PF16_REG(I)                                                                                                   \
            {                                                                                                            \
                const float ss = sc_v[I];                                                                                \
                const uint32_t row = sub_row0 + (uint32_t)(8 * ((I) >> 2) + ((I) & 3));                                    \
                bool pass = ss > tau_s;                                                                                  \
                if (check_rows) pass = pass && ((int64_t)row < row_end);                                                 \
                if (pass) {                                                                                              \
                    if (ccnt < (uint32_t)PF2_CAND) {                                                                     \
                        const float s_ = ss * down;                      /* the approximate score itself (exact: a power of two) */ \
                        if (hg.counters != nullptr) ms_hist_count(hg, s_);                                               \
                        *cand_slot(ccnt) = ms_u32x2{__float_as_uint(s_), row};                                           \
                        ccnt += 1;                                                                                       \
                        sc_v[I] = -INFINITY;                                                                             \
                    } else {                                                                                             \
                        ovf = true;                                                                                      \
                    }                                                                                                    \
                }                                                                                                        \
            }
#define MS_PF16_GROUP(G, GM) if (__ballot(GM > tau_s) != 0) { MS_PF16_REG(4 * (G)) MS_PF16_REG(4 * (G) + 1) MS_PF16_REG(4 * (G) + 2) MS_PF16_REG(4 * (G) + 3) }
            MS_PF16_GROUP(0, g0) MS_PF16_GROUP(1, g1) MS_PF16_GROUP(2, g2) MS_PF16_GROUP(3, g3)
#undef MS_PF16_GROUP
#undef MS_PF16_REG
            if (__builtin_expect(__ballot(ovf) == 0, 1)) break;
            flush();                                        // some lane's buffer was full: the lists take the buffers, then the rest of the tile
        }
    };
    auto visit_v1 = [&](f32x16 &sc_v, int t, int half, bool check_rows) __attribute__((always_inline)) {
        if (mask_on) apply_mask(sc_v, t, half);
        const uint32_t sub_row0 = (uint32_t)(row_begin + (int64_t)t * 64) + (uint32_t)(32 * half + 4 * h);
        uint32_t regs = 0;                                   // registers holding a candidate of some lane (uniform)
#if MS_PF16_GROUPED
        // Round 6: sixteen ballots (~64 instructions, candidate or not) were half of a typical visit.  The lane's maximum over each GROUP of
        // four registers (rows 8 g + 4 h + 0..3: eight instructions), one ballot per group, and the four ballots of a group only where its
        // maximum passes: 8 + 12 + 16 instructions for the usual visit (one register of one group).
        {
            float g0, g1, g2, g3;
            asm("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %8, %9, %10\n\tv_max3_f32 %2, %12, %13, %14\n\tv_max3_f32 %3, %16, %17, %18\n\t"
                "v_max_f32 %0, %0, %7\n\tv_max_f32 %1, %1, %11\n\tv_max_f32 %2, %2, %15\n\tv_max_f32 %3, %3, %19"
                : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
                : "v"(sc_v[0]), "v"(sc_v[1]), "v"(sc_v[2]), "v"(sc_v[3]), "v"(sc_v[4]), "v"(sc_v[5]), "v"(sc_v[6]), "v"(sc_v[7]), "v"(sc_v[8]),
                  "v"(sc_v[9]), "v"(sc_v[10]), "v"(sc_v[11]), "v"(sc_v[12]), "v"(sc_v[13]), "v"(sc_v[14]), "v"(sc_v[15]));
#define MS_PF16_GROUP(G, GM)                                                                                             \
            if (__ballot(GM > tau_s) != 0) {                                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) regs |= (__ballot(sc_v[4 * (G) + j] > tau_s) != 0 ? 1u : 0u) << (4 * (G) + j); \
            }
            MS_PF16_GROUP(0, g0) MS_PF16_GROUP(1, g1) MS_PF16_GROUP(2, g2) MS_PF16_GROUP(3, g3)
#undef MS_PF16_GROUP
        }
#else
#pragma unroll
        for (int i = 0; i < 16; ++i) regs |= (__ballot(sc_v[i] > tau_s) != 0 ? 1u : 0u) << i;
#endif
#pragma unroll 1
        while (regs != 0u) {
            const int i = __builtin_ctz(regs);
            regs &= regs - 1u;
            // register i, i uniform: a select tree of 15 v_cndmask under scalar masks (written as asm: left to itself hipcc turns
            // any such selection into a dynamic index through scratch memory)
            const uint64_t m0_ = (i & 1) ? ~0ull : 0ull, m1_ = (i & 2) ? ~0ull : 0ull, m2_ = (i & 4) ? ~0ull : 0ull, m3_ = (i & 8) ? ~0ull : 0ull;
            float l1[8], l2[4], l3[2], ss;
#define MS_SEL(D, A, B, M) asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "s"(M))
#pragma unroll
            for (int j = 0; j < 8; ++j) MS_SEL(l1[j], sc_v[2 * j], sc_v[2 * j + 1], m0_);
#pragma unroll
            for (int j = 0; j < 4; ++j) MS_SEL(l2[j], l1[2 * j], l1[2 * j + 1], m1_);
            MS_SEL(l3[0], l2[0], l2[1], m2_); MS_SEL(l3[1], l2[2], l2[3], m2_);
            MS_SEL(ss, l3[0], l3[1], m3_);
#undef MS_SEL
            const uint32_t row = sub_row0 + (uint32_t)(8 * (i >> 2) + (i & 3));
            const float s = ss * down;                      // the approximate score itself (exact: a power of two)
            bool pass = ss > tau_s;
            if (check_rows) pass = pass && ((int64_t)row < row_end);
            if (pass) {
                if (hg.counters != nullptr) ms_hist_count(hg, s);
                *cand_slot(ccnt) = ms_u32x2{__float_as_uint(s), row};
                ccnt += 1;
            }
            if (__builtin_expect(__ballot(ccnt >= PF2_CAND) != 0, 0)) flush();     // some lane's buffer is full
        }
    };
    auto visit = [&](f32x16 &sc_v, int t, int half, bool check_rows) __attribute__((always_inline)) {
        if constexpr (VISIT2) visit_v2(sc_v, t, half, check_rows); else visit_v1(sc_v, t, half, check_rows);
    };

#ifdef MS_STAMP
    unsigned long long sp_t0 = 0, sp_sync = 0, sp_vis = 0, sp_nvis = 0, sp_chain = 0, sp_hist = 0, sp_lgkm = 0, sp_flow = 0;
#define PF2_T0() sp_t0 = __builtin_amdgcn_s_memtime()
#define PF2_ACC(V) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); V += now_ - sp_t0; sp_t0 = now_; }
#else
#define PF2_T0()
#define PF2_ACC(V)
#endif
    // ---- pipeline.  Stage t: the fragments of tile t are in `fr`; each is replaced by tile t + 1's right behind the matrix
    //      instructions that used it; the chain runs into `out`; the scores of tile t - 1 (`pv`) are filtered in its shadow, the
    //      rare path follows.  Two stages per loop iteration swap (pv, out): no register copies.
    f32x4 fr[16];
    f32x16 accA0, accA1, accB0, accB1;         // (A, B) = (previous, current) tile; 0 / 1 = rows 0-31 / 32-63 of it
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA0[i] = -INFINITY; accA1[i] = -INFINITY; accB0[i] = -INFINITY; accB1[i] = -INFINITY; }

    auto frag_base = [&](int t) __attribute__((always_inline)) -> const f32x4 * {
        return reinterpret_cast<const f32x4 *>(smem + (size_t)(t % PF2_R) * 16384 + 16 * lane);
    };
    // filter of tile t - 1 (its scores in pv0 / pv1): one compare per tile; the rare path only where a lane's maximum passes
    auto filter = [&](int t, f32x16 &pv0, f32x16 &pv1) __attribute__((always_inline)) {
        if (SAMPLE) {
            if (mask_on && t > 0) { apply_mask(pv0, t - 1, 0); apply_mask(pv1, t - 1, 1); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { smax = (pv0[i] > smax) ? pv0[i] : smax; smax = (pv1[i] > smax) ? pv1[i] : smax; }       // (NaN scores never enter)
        } else {
            // (eight instructions; `fmaxf` compiles to ten: hipcc canonicalises the first two operands.  v_max3 ignores NaNs as fmaxf does)
            // (ONE statement: hipcc pads every asm statement with an s_nop)
            float mx, mx1;
#define MS_PF16_MAX(M, PV)                                                                                                                        \
            asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"             \
                "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"               \
                : "=&v"(M) : "v"(PV[0]), "v"(PV[1]), "v"(PV[2]), "v"(PV[3]), "v"(PV[4]), "v"(PV[5]), "v"(PV[6]), "v"(PV[7]), "v"(PV[8]), "v"(PV[9]), \
                  "v"(PV[10]), "v"(PV[11]), "v"(PV[12]), "v"(PV[13]), "v"(PV[14]), "v"(PV[15]))
#ifdef MS_PF16_ABL_NOFILTER
            asm volatile("" : "+v"(pv0), "+v"(pv1));
            mx = -INFINITY; mx1 = -INFINITY;
#else
            MS_PF16_MAX(mx, pv0);
            MS_PF16_MAX(mx1, pv1);
#endif
#undef MS_PF16_MAX
            // (each half tile is visited only if one of ITS scores passes: a visit costs ~900 cycles per half, candidate or not)
            const bool hit0 = __ballot(mx > tau_s) != 0, hit1 = __ballot(mx1 > tau_s) != 0;
#ifdef MS_PF2_NOVISIT
            asm volatile("" ::"v"(mx));
            if (false) {
#else
            if (__builtin_expect(hit0 || hit1 || neg_tau, 0)) {
#endif
                PF2_T0();
                if (t > 0) {
                    if (hit0 || neg_tau) visit(pv0, t - 1, 0, false);
                    if (hit1 || neg_tau) visit(pv1, t - 1, 1, false);
                }
                if (mask_on) neg_tau = __ballot(st.tau < 0.0f) != 0;
#ifdef MS_STAMP
                sp_nvis += 1;
#endif
                PF2_ACC(sp_vis)
            }
        }
    };
    // STEADY: 2 <= t and t + D < ntl -- every condition of the head and the tail of a stream is known (the generic form is the
    // same code with the tests in)
    auto stage = [&](auto steady_c, int t, f32x16 &pv0, f32x16 &pv1, f32x16 &out0, f32x16 &out1) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        // tile t is in registers (and the snapshot of the counters taken during the last chain): its slot is free
        PF2_T0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PF2_ACC(sp_lgkm)
        if (STEADY || t + 1 < ntl) wait_arrived(t + 1);                   // tile t + 1 has landed from every wave
        PF2_ACC(sp_sync)
        // MS_PF16_SHADOW == 0: ONE pair of accumulators -- the scores of tile t - 1 are filtered here, before the chain of tile t
        // overwrites them (the partner wave of the SIMD has the matrix pipe meanwhile); == 1: in the shadow of the chain, from a
        // second pair (32 more registers: the 10- and 16-entry lists then spill with the query split hi / lo)
        if constexpr (!MS_PF16_SHADOW) filter(t, pv0, pv1);
#pragma unroll
        for (int i = 0; i < 16; ++i) { out0[i] = 0.0f; out1[i] = 0.0f; }
        // The chain, k block by k block; behind each block's matrix instructions the two fragment reads of tile t + 1 that replace
        // its operands (past the last tile: a stale slot, never used) and ONE of everything else -- an LDS-DMA piece costs the wave
        // 8 cycles next to a matrix instruction and 60-185 in a burst.
        const f32x4 *src = frag_base(t + 1);
        const bool issuing = STEADY || t + PF2_D < ntl;         // (uniform)
#ifdef MS_PF16_ABL_NOMFMA
#define MS_PF16_MFMA(ACC, F, Q) asm volatile("" : "+v"(ACC) : "v"(F), "v"(Q));
#elif defined(MS_PF16_ABL_BF16)
// (diagnostic build, WRONG results: the same bits through the bf16 form of the instruction -- does the narrower multiplier hold a higher clock?)
#define MS_PF16_MFMA(ACC, F, Q) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F), __builtin_bit_cast(bf16x8, Q), ACC, 0, 0, 0);
#else
#define MS_PF16_MFMA(ACC, F, Q) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(F, Q, ACC, 0, 0, 0);
#endif
#define MS_PF2_BLOCK(B)                                                                                               \
        {                                                                                                             \
            const f16x8 f0 = __builtin_bit_cast(f16x8, fr[2 * (B)]), f1 = __builtin_bit_cast(f16x8, fr[2 * (B) + 1]);   \
            MS_PF16_MFMA(out0, f0, qh[B])                                                                             \
            MS_PF16_MFMA(out1, f1, qh[B])                                                                             \
            if constexpr (NQP == 2) {                                                                                 \
                out0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, ql[B], out0, 0, 0, 0);                              \
                out1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, ql[B], out1, 0, 0, 0);                              \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            fr[2 * (B)] = src[64 * (2 * (B))];                                                                        \
            MS_PF16_SECOND_READ(B)                                                                                    \
        }
#if MS_PF16_ILV
        // (MS_PF16_ILV: matrix instruction, the read that replaces ITS operand, matrix instruction, read -- each read issues in the 32-cycle
        //  shadow of the instruction in front of it, and only "everything else" follows the second one)
#undef MS_PF2_BLOCK
#define MS_PF2_BLOCK(B)                                                                                               \
        {                                                                                                             \
            const f16x8 f0 = __builtin_bit_cast(f16x8, fr[2 * (B)]), f1 = __builtin_bit_cast(f16x8, fr[2 * (B) + 1]);   \
            MS_PF16_MFMA(out0, f0, qh[B])                                                                             \
            if constexpr (NQP == 2) out0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, ql[B], out0, 0, 0, 0);          \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            fr[2 * (B)] = src[64 * (2 * (B))];                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            MS_PF16_MFMA(out1, f1, qh[B])                                                                             \
            if constexpr (NQP == 2) out1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, ql[B], out1, 0, 0, 0);          \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            MS_PF16_SECOND_READ(B)                                                                                    \
        }
#endif
#ifdef MS_PF16_ABL_1616
        // (diagnostic build, WRONG results: the same operands through 32 v_mfma_f32_16x16x32_f16 on eight 4-register accumulators -- what would the
        //  other matrix shape's instruction mix run at?)
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        f32x4_ oa[4], ob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { oa[i] = f32x4_{0.0f, 0.0f, 0.0f, 0.0f}; ob[i] = f32x4_{0.0f, 0.0f, 0.0f, 0.0f}; }
#undef MS_PF2_BLOCK
#define MS_PF2_BLOCK(B)                                                                                               \
        {                                                                                                             \
            const f16x8 f0 = __builtin_bit_cast(f16x8, fr[2 * (B)]), f1 = __builtin_bit_cast(f16x8, fr[2 * (B) + 1]);   \
            oa[(B) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, qh[B], oa[(B) & 3], 0, 0, 0);                    \
            ob[(B) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, qh[B], ob[(B) & 3], 0, 0, 0);                    \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            fr[2 * (B)] = src[64 * (2 * (B))];                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            oa[((B) + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f0, qh[((B) + 1) & 7], oa[((B) + 2) & 3], 0, 0, 0); \
            ob[((B) + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1, qh[((B) + 1) & 7], ob[((B) + 2) & 3], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            MS_PF16_SECOND_READ(B)                                                                                    \
        }
#endif
        MS_PF2_BLOCK(0)
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(1)
        if (STEADY || issuing) issue_prep(t + PF2_D);           // (slot (t + D) % R is free: see the ring geometry)
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(2)
        if (STEADY || issuing) issue_piece(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(3)
        if (STEADY || issuing) issue_piece(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(4)
        if (STEADY || issuing) issue_piece(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(5)
        if (STEADY || issuing) { issue_piece(std::integral_constant<int, 3>{}); issue_aux(t + PF2_D); }
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(6)
        // own pieces of tile t + W have landed (issued D - W stages ago; the tail of a stream drains): one more arrival for that tile
        PF2_ACC(sp_chain)
        if (STEADY || issuing) wait_own(std::integral_constant<int, PF2_D - PF2_W>{}); else ms_pf2_vmcnt<0>();
        PF2_ACC(sp_flow)         // (diagnostic builds: cycles stalled on this wave's own pieces, inside the chain)
        if (STEADY || t + PF2_W < ntl) publish(t + PF2_W);
        read_arrived(t + 2);                                    // for the next stage (the other waves publish during their chains)
        __builtin_amdgcn_sched_barrier(0);
        MS_PF2_BLOCK(7)
#undef MS_PF2_BLOCK
#ifdef MS_PF16_ABL_1616
#pragma unroll
        for (int i = 0; i < 16; ++i) { out0[i] = oa[i >> 2][i & 3]; out1[i] = ob[i >> 2][i & 3]; }
#endif
#ifdef MS_STAMP
        asm volatile("s_nop 0" : "+v"(out0), "+v"(out1));
#endif
        PF2_ACC(sp_chain)
        if constexpr (MS_PF16_SHADOW) filter(t, pv0, pv1);
    };

#ifdef MS_STAMP
    const unsigned long long sp_c0 = __builtin_amdgcn_s_memtime(), sp_r0 = __builtin_amdgcn_s_memrealtime();
    tl_setup = sp_r0;
#endif
    if (ntl > 0) {
        // the first W tiles: own pieces, then (tile 0) everybody's.  With fewer than D tiles fewer pieces were issued: drain.
        if (ntl >= PF2_D) wait_own(std::integral_constant<int, PF2_D - PF2_W>{}); else ms_pf2_vmcnt<0>();
#pragma unroll
        for (int t = 0; t < PF2_W; ++t)
            if (t < ntl) publish(t);
        read_arrived(0);
        wait_arrived(0);
#ifdef MS_STAMP
        tl_first = __builtin_amdgcn_s_memrealtime();
#endif
        {
            const f32x4 *src = frag_base(0);
#pragma unroll
            for (int f = 0; f < 16; ++f) fr[f] = src[64 * f];
        }
        read_arrived(1);
        // Shared bound, every MS_HIST_PERIOD tiles: the 16 bucket counters of this wave's 32 queries are fetched by LDS-DMA (sc1:
        // past this CU's L1) -- no destination register, nothing the compiler has to wait for; the counted vector-memory waits of
        // the stages cover them -- and read back one iteration later: the highest bucket edge with at least k rows at or above
        // it (counted by all waves so far) bounds the k-th best.  Waves w and w + 4 share staging area w & 3: wave w fetches in
        // phase 2 w of a period and reads in phase 2 w + 2, so the two are half a period (>= 8 tiles) apart, and no wave runs
        // more than max(W - 1, R - D) tiles ahead of another.
        // (round 6, MS_PF2_HIST_AREAS = 8: every wave has a staging area of its own and the period may be 8 tiles -- the first look at the
        //  shared bound comes at tile 2..8 instead of 2..16 and every 8 tiles from then on: C2's 61-tile streams see it 7 times, not 3-4)
        constexpr int HP = MS_PF16_HIST_PERIOD;
        static_assert(HP == 16 || (HP == 8 && (PF2_HIST_AREAS == 8 || NW == 4)), "a shared staging area needs half a period of 16 tiles between its two users");
        const int fetch_phase = (HP == 16 ? 2 * wave : 2 * (wave & 3)), read_phase = (fetch_phase + 2) & (HP - 1);
        const int hist_area = wave & (PF2_HIST_AREAS - 1);
        auto hist_step = [&](int t) __attribute__((always_inline)) {
            const int phase = t & (HP - 1);
            if (phase == fetch_phase) {
                const uint64_t hb = (uint64_t)(uintptr_t)p.hist + (uint64_t)qtile * 2048u;
                const uint32_t hb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)hb);
                const uint32_t hb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
                const uint64_t shb = ((uint64_t)hb_hi << 32) | (uint64_t)hb_lo;
                const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(ring_lds + PF2_OFF_HIST + (uint32_t)hist_area * 2048u);
                ms_glds_s16_sc1<0>(dst, voff, shb);
                ms_glds_s16_sc1<1024>(dst + 1024, voff, shb);
            }
            if (phase == read_phase && t >= 2) {
                PF2_T0();
                // the two stages since then issued two tiles' pieces behind the counters' (near the end of a stream: fewer -- drain)
                if (t - 1 + PF2_D < ntl) wait_own_hist(); else ms_pf2_vmcnt<0>();
                const ms_u32x4 *hp = reinterpret_cast<const ms_u32x4 *>(smem + PF2_OFF_HIST + hist_area * 2048 + r * 64);
                const ms_u32x4 c0 = hp[0], c1 = hp[1], c2 = hp[2], c3 = hp[3];
                const uint32_t c[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
                uint32_t cum = 0;
                int n_lt = 0;
#pragma unroll
                for (int j = 15; j >= 0; --j) { cum += c[j]; n_lt += (cum < (uint32_t)p.k) ? 1 : 0; }
                const int J = 15 - n_lt;
                if (hg.counters != nullptr && J >= 1) {
                    st.floor = fmaxf(st.floor, ms_next_below(ms_hist_edge(hg, J)));
                    st.tau = fmaxf(st.tau, st.floor);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the staging area has been read)
                retau();
                if (mask_on) neg_tau = __ballot(st.tau < 0.0f) != 0;
                PF2_ACC(sp_hist)
            }
        };
        int t = 0;
#if !MS_PF16_SHADOW
        // ONE stage per iteration (round 6: two per iteration doubled every copy of the rare path in the instruction cache; the register swap
        // that needed pairs belongs to the shadow form): the body of a stream -- every stage issues (t + D < ntl) -- ...
        for (; t + PF2_D < ntl; ++t) {
            if ((t & 1) == 0) {
                if (pace_on) pace(t);
                if (hist_on) hist_step(t);
            }
            stage(std::true_type{}, t, accA0, accA1, accA0, accA1);
        }
        // ... and its last D tiles, with the tests in
        for (; t < ntl; ++t) {
            if ((t & 1) == 0 && hist_on) hist_step(t);
            stage(std::false_type{}, t, accA0, accA1, accA0, accA1);
        }
#else
        // the body of a stream: every stage issues (t + 1 + D < ntl), two tiles per iteration ...
        for (; t + 1 + PF2_D < ntl; t += 2) {
            if (pace_on) pace(t);
            if (hist_on) hist_step(t);
            if constexpr (MS_PF16_SHADOW) {
                stage(std::true_type{}, t, accA0, accA1, accB0, accB1);             // accA = scores of tile t - 1 (or -inf), accB <- tile t
                stage(std::true_type{}, t + 1, accB0, accB1, accA0, accA1);         // accB = tile t, accA <- tile t + 1
            } else {
                stage(std::true_type{}, t, accA0, accA1, accA0, accA1);
                stage(std::true_type{}, t + 1, accA0, accA1, accA0, accA1);
            }
        }
        // ... and its last D + 1 tiles, with the tests in
        for (; t + 1 < ntl; t += 2) {
            if (hist_on) hist_step(t);
            if constexpr (MS_PF16_SHADOW) {
                stage(std::false_type{}, t, accA0, accA1, accB0, accB1);
                stage(std::false_type{}, t + 1, accB0, accB1, accA0, accA1);
            } else {
                stage(std::false_type{}, t, accA0, accA1, accA0, accA1);
                stage(std::false_type{}, t + 1, accA0, accA1, accA0, accA1);
            }
        }
        if (t < ntl) {
            if constexpr (MS_PF16_SHADOW) {
                stage(std::false_type{}, t, accA0, accA1, accB0, accB1);
                accA0 = accB0; accA1 = accB1;
            } else {
                stage(std::false_type{}, t, accA0, accA1, accA0, accA1);
            }
        }
#endif
#ifdef MS_STAMP
        if (!SAMPLE && lane == 0 && p.stamps != nullptr && (size_t)bid * 64 + 64 <= 4 * 4 * 65536) {
            unsigned long long *o = p.stamps + ((size_t)bid * 8 + wave) * 8;
            tl_loop = __builtin_amdgcn_s_memrealtime();
            o[0] = __builtin_amdgcn_s_memtime() - sp_c0;
            o[1] = __builtin_amdgcn_s_memrealtime() - sp_r0;
            o[2] = (unsigned long long)ntl;
            o[3] = (sp_lgkm << 32) | (sp_flow & 0xFFFFFFFFull);
            o[4] = sp_vis; o[5] = sp_nvis; o[6] = sp_chain; o[7] = (sp_hist << 32) | (sp_sync & 0xFFFFFFFFull);
        }
#endif
        if (pace_on && wave == 0 && lane == 0)        // this workgroup is through: nobody waits for it any more
            __hip_atomic_store(p.prog + (size_t)stream * 16 + qg, (p.prog_epoch << 24) | 0xFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // drain: the last tile's scores (accA), rows past row_end rejected
        if (SAMPLE) {
            if (mask_on) { apply_mask(accA0, ntl - 1, 0); apply_mask(accA1, ntl - 1, 1); }
#pragma unroll
            for (int i = 0; i < 16; ++i) { smax = (accA0[i] > smax) ? accA0[i] : smax; smax = (accA1[i] > smax) ? accA1[i] : smax; }
        } else {
            visit(accA0, ntl - 1, 0, true);
            visit(accA1, ntl - 1, 1, true);
            flush();
        }
    }

    if (SAMPLE) {
        // the stream's entry: the two half-tile maxima of the lane pair (scores of distinct rows), larger first; rows are
        // not recorded (the bound selection reads values only), a distinct placeholder keeps the slots "occupied"
        smax = smax * down;                                     // (-inf stays -inf)
        const float other = ms_xor32_f(smax, h);
        if (h == 0) {
            const float hi = (other > smax) ? other : smax, lo = (other > smax) ? smax : other;
            const size_t o = ((size_t)stream * p.nq_pad + qidx) * p.k;          // stream-major lists (below)
            p.part_s[o] = hi;
            p.part_i[o] = (hi > -INFINITY) ? (uint32_t)(2 * stream) : MS_IDX_NONE;
            if (p.k > 1) {
                p.part_s[o + 1] = lo;
                p.part_i[o + 1] = (lo > -INFINITY) ? (uint32_t)(2 * stream + 1) : MS_IDX_NONE;
            }
        }
        return;
    }
#ifdef MS_PF16_NOWRITE          // (diagnostic build: what do the list stores cost?)
    if (p.k > 0) return;
#endif
#pragma unroll
    for (int j = 0; j < (SAMPLE ? 1 : KL); ++j) {
        const int rank = h * KL + j;
        if (rank < p.k) {
            // STREAM-MAJOR lists, [stream][query][rank] (round 5): a workgroup's lists are ONE contiguous block (a wave's 32 queries:
            // 32 k entries back to back), which the L2 merges into whole lines before they leave for HBM.  The rank-major layout of
            // the fp32 scans ([query][rank][stream]) put every 4-byte entry of this kernel on a line of its own, shared with 255
            // other workgroups on other XCDs: 118 MB of HBM writes for a 10 MB payload at C2, ~25 us of the launch
            // (profiles/r05_pf_list_layout_ab.log).  ms_sample_bound_kernel and ms_block_merge_kernel read it (`sm_stride`).
            const size_t o = ((size_t)stream * p.nq_pad + qidx) * p.k + rank;
            p.part_s[o] = st.ls[j];
            p.part_i[o] = st.li[j];
        }
    }
#ifdef MS_STAMP
    if (!SAMPLE && lane == 0 && p.stamps != nullptr && bid < 4096) {      // timeline of this wave (100 MHz ticks, absolute): second half of the buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *o = p.stamps + 8 * 8 * 4096 + ((size_t)bid * 8 + wave) * 8;
        o[0] = tl_entry; o[1] = tl_setup; o[2] = tl_first; o[3] = tl_loop; o[4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int KL, int NW, bool MASK, int NQP>
int launch_scan_pf16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_pf16_kernel<KL, NW, false, MASK, NQP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF2_LDS));
    hipLaunchKernelGGL((ms_scan_pf16_kernel<KL, NW, false, MASK, NQP>), dim3(pl.grid), dim3(64 * NW), PF2_LDS, st, sp);
    MS_LAUNCH_CHECK("ms_scan_pf16_kernel");
    return MS_OK;
}
template <int KL, int NW>
int launch_scan_pf16_any(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    const bool two = sp.pf_format == MS_PF_F16X2;
    if (sp.lengths != nullptr) return two ? launch_scan_pf16<KL, NW, true, 2>(pl, sp, st) : launch_scan_pf16<KL, NW, true, 1>(pl, sp, st);
    return two ? launch_scan_pf16<KL, NW, false, 2>(pl, sp, st) : launch_scan_pf16<KL, NW, false, 1>(pl, sp, st);
}

// One non-template entry point per list length (ms_scan_pf16_kl*.hip); the sample pass and the image builder live in ms_scan_pf16_kl5.hip.
int ms_launch_scan_pf16_kl5(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_pf16_kl10(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_pf16_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_pf16_kl32(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_sample_pf16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_pf16_build_image(const float *db, int64_t n, int sr, void *image, hipStream_t st);
