// The prefilter's scan over the SPLIT IMAGE of the database (ms_pf_build_image): the rows' bf16 hi / lo halves laid out once, when
// the database becomes resident, in the order the matrix instruction wants them -- so that the scan issues no conversion at all
// (round 3's prefilter split every float in registers, in every one of the four compute waves of a workgroup: ~190 vector
// instructions per 24 matrix instructions, 27 % of the split-bf16 matrix roof).
//
// Image: tile T (rows 32 T .. 32 T + 31, zero rows past the end) = 16 KiB at byte 16384 T; inside it fragment f = 2 b + part
// (k block b = 0..7, part 0 = hi, 1 = lo) = 1 KiB at 1024 f, lane (r, h) = 16 bytes at 16 (32 h + r): the eight bf16 of row r at
// dimensions 64 h + 8 b + j, j = 0..7.  512 B per row, like the fp32 rows -- which stay resident for the exact re-scoring.
//   * a tile arrives by SIXTEEN linear 1 KiB LDS-DMA pieces (perfectly coalesced), no swizzle, no gather;
//   * lane (r, h) reads fragment f at slot + 16 lane + 1024 f: one ds_read_b128 per fragment, conflict free, immediate offsets;
//   * per tile and query tile: 24 v_mfma_f32_32x32x16_bf16 (hi.qhi, hi.qlo, lo.qhi per k block) = 768 cycles of matrix pipe.
//
// Workgroup = NW waves (8: two per SIMD, <= 256 registers each; 4 for up to four query tiles and for 32-entry lists), one query
// tile per wave, NO loader wave: every wave loads 16 / NW pieces of every tile D tiles ahead into a ring of R slots shared by
// the workgroup and scans every tile against its query tile: 8 query tiles share one copy of the rows in LDS (round 3: 4), which
// halves the LDS-DMA and L2 traffic per query and lets C2's 256 queries run as ONE query group.  Two waves per SIMD is what
// hides everything that is not a matrix instruction -- synchronisation, filters, the rare path: a wave alone on its SIMD issues
// every instruction at ~5 cycles with the matrix pipe idle meanwhile (first version of this kernel, 4 waves x 2 query tiles:
// 2,670 cycles per tile for 1,536 of matrix work even with the rare path compiled out; stamps in DESIGN.md).
// No barrier inside the scan: the waves synchronise through one ARRIVAL COUNTER per tile in LDS (16 of them, reused modulo 16): a
// wave whose pieces of tile t + W have landed (a counted vector-memory wait, D - W stages after it issued them) adds one to that
// tile's counter; a wave starts stage t when the counter of tile t + 1 shows NW arrivals per use -- one ds_read_b32 taken during the
// previous stage's chain, one scalar compare.  A wave may run W - 1 tiles ahead of the slowest one.
//
// The rare path is NOT the loader-wave kernel's.  A visit only APPENDS: a lane whose score passes its threshold counts it in the
// shared bound's histogram and stores (score, row) into a buffer of its own in LDS (PF2_CAND entries; ~150 instructions per
// visit instead of ~600 for "ballots, row masks, sorted insertion"); the sorted lists in registers take the buffered candidates
// at the end of the stream -- or when some lane's buffer is full -- PF2_CAND rounds of one insertion step for all 64 lanes at
// once.  Thresholds therefore come from the sample pass and the shared bound (which is where they came from anyway: a stream's
// own list hardly ever beats them).  The lists are filled in no particular row order: ties in APPROXIMATE score are kept in no
// particular order, which is all the exact re-scoring needs.  The code is kept compact on purpose (the kernel has to stay well
// inside the 64 KiB instruction cache two CUs share: a version with the append steps unrolled was 96 KiB and paid ~3,000 cycles
// of instruction fetch per visit).
// Otherwise as there (ms_scan.h): lists in the registers of the query's two lanes, one compare per tile in front of the rare
// path, the sample pass (SAMPLE), the shared bound's counting histogram, the output format.
// Cosine on unit rows (MS_MODE_COSINE_UNIT, p.lengths != NULL): the length mask is applied in the rare path (and to every score
// of the sample pass); a wave with a negative threshold visits the rare path for every tile, as AUXM == 2 does there.
#pragma once
#include "ms_scan.h"

#ifndef MS_PF2_R
#define MS_PF2_R 8
#endif
#ifndef MS_PF2_D
#define MS_PF2_D 6
#endif
#ifndef MS_PF2_W
#define MS_PF2_W 3
#endif
constexpr int PF2_R = MS_PF2_R;                // ring slots (16 KiB tiles)
constexpr int PF2_D = MS_PF2_D;                // a wave issues its pieces of tile t + D during stage t
constexpr int PF2_W = MS_PF2_W;                // ... then waits for its pieces of tiles <= t + W (issued D - W stages ago: HBM latency) and
                                               // publishes them
// No "consumed" counters: no wave starts stage t before every wave has published tile t + 1, which a wave does near the end of its
// stage t + 1 - W, with tile t + 1 - W in its registers and all but the last two fragments of tile t + 2 - W read.  So when a wave
// issues tile t + D during stage t, every wave has pulled tiles <= t + 1 - W into registers, and the slot of tile t + D - R is free
// as long as R >= D + W - 1.  A wave may run W - 1 tiles ahead of the slowest one; its pieces have D - W stages to arrive before
// it waits for them.
static_assert(PF2_W >= 2 && PF2_D - PF2_W >= 1 && PF2_R >= PF2_D + PF2_W - 1, "ring geometry");
constexpr int PF2_AUXR = 16;                   // aux ring: the row lengths of a tile (256 B per slot), cosine mode
constexpr int PF2_CAND = 4;                    // candidates a lane buffers before the lists take them
constexpr int PF2_OFF_AUX = PF2_R * 16384;
#ifndef MS_PF2_HIST_AREAS
#define MS_PF2_HIST_AREAS 4
#endif
constexpr int PF2_HIST_AREAS = MS_PF2_HIST_AREAS;               // 4: waves w and w + 4 take turns at area w & 3 (half a period of 16 tiles apart); 8: one per wave
static_assert(PF2_HIST_AREAS == 4 || PF2_HIST_AREAS == 8, "staging areas of the shared bound");
constexpr int PF2_OFF_HIST = PF2_OFF_AUX + PF2_AUXR * 256;      // the shared bound's counters of a wave's 32 queries, staged by LDS-DMA: 2 KiB per area
constexpr int PF2_OFF_CAND = PF2_OFF_HIST + PF2_HIST_AREAS * 2048;           // [wave][slot][lane] (score, row): 512 B per slot
constexpr int PF2_ARR = 16;                                     // arrival counters: one per tile modulo 16, counting up by NW per reuse
constexpr int PF2_OFF_CNT = PF2_OFF_CAND + 8 * PF2_CAND * 512;  // arrived[16]
constexpr int PF2_OFF_DUMMY = PF2_OFF_CNT + 64;                 // cosine mode: where the waves other than 0 drop their (unused) aux piece
constexpr int PF2_LDS = PF2_OFF_DUMMY + 7 * 256;
static_assert(PF2_ARR > PF2_R + PF2_W, "a counter is not reused while anybody may still wait for its previous tile");
static_assert(PF2_LDS <= 160 * 1024, "LDS of one CU");
static_assert(MS_HIST_PERIOD >= 16, "eight waves take turns at four staging areas: phases 2 w and 2 w + 2 of a period");

typedef uint32_t ms_u32x4 __attribute__((ext_vector_type(4)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int N>
__device__ __forceinline__ void ms_pf2_vmcnt() {        // (inline asm: the compiler does not know about the LDS-DMA pieces in flight)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
#pragma clang diagnostic pop

// The approximate score of (row, query): ONE accumulator chain, k blocks in order, per block hi.qhi, hi.qlo, lo.qhi.  The sample
// pass and the full pass run exactly this sequence (the sample's bound must hold bit for bit).
// MASK: MS_MODE_COSINE_UNIT with a length mask (p.lengths != NULL).
template <int KL, int NW, bool SAMPLE, bool MASK>
__global__ __launch_bounds__(64 * NW, NW / 4) void ms_scan_pf2_kernel(const ScanParams p) {
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    constexpr int PPW = 16 / NW;                       // LDS-DMA pieces of a tile per wave
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
    const int nfull = (int)((row_end - row_begin) >> 5);
    const int rem = (int)((row_end - row_begin) & 31);
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
    const uint64_t img0 = (uint64_t)(uintptr_t)p.pf_image + (uint64_t)(row_begin >> 5) * 16384u + (uint32_t)(1024 * PPW) * (uint32_t)wave;
    // (cosine mode: EVERY wave issues one more piece per tile so that the counted waits are the same for all of them; only wave 0's
    //  -- the rows' lengths -- is read)
    uint64_t it_sb = 0;            // base address and LDS destination of the tile being issued (uniform)
    uint32_t it_dst = 0;
    auto issue_prep = [&](int t) __attribute__((always_inline)) {
        const uint64_t b = img0 + (uint64_t)t * 16384u;
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        it_sb = ((uint64_t)b_hi << 32) | (uint64_t)b_lo;
        it_dst = (uint32_t)__builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)(t % PF2_R) * 16384u + (uint32_t)(1024 * PPW) * (uint32_t)wave);
    };
    auto issue_piece = [&](auto i_c) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value;
        if constexpr (I < PPW) {
            // (uniform values that live across branches: say so again, or the "s" operands of the asm may be handed vector registers)
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane(it_dst);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)it_sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(it_sb >> 32));
#ifdef MS_PF2_NT
            ms_glds_s16_nt<1024 * I>(d + 1024 * I, voff, ((uint64_t)hi << 32) | (uint64_t)lo);
#else
            ms_glds_s16<1024 * I>(d + 1024 * I, voff, ((uint64_t)hi << 32) | (uint64_t)lo);
#endif
        }
    };
    auto issue_aux = [&](int t) __attribute__((always_inline)) {
        if constexpr (MASK) {
            int64_t row = row_begin + (int64_t)t * 32 + r;
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
    // own pieces of every tile but the youngest N issued have landed
    auto wait_own = [&](auto n_c) __attribute__((always_inline)) {
        constexpr int N = decltype(n_c)::value;
        ms_pf2_vmcnt<(PPW + (MASK ? 1 : 0)) * N>();
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
                     "s_cbranch_scc1 2f\n\t"
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
    // ---- prologue: the first D tiles are requested before anything else (HBM latency overlaps the query set-up)
#pragma unroll
    for (int t = 0; t < PF2_D; ++t)
        if (t < ntl) issue_tile(t);

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
    bf16x8 qhi[8], qlo[8];
    const int qidx = qtile * 32 + r;
    const bool q_valid = qidx < p.nq;
    const bool hist_on = !SAMPLE && p.hist != nullptr && p.lb_s != nullptr;      // (uniform)
#pragma unroll
    for (int j = 0; j < (SAMPLE ? 1 : KL); ++j) { st.ls[j] = -INFINITY; st.li[j] = MS_IDX_NONE; }
    st.floor = -INFINITY;
    st.tau = -INFINITY;
    if (!SAMPLE && p.lb_s != nullptr) {
        const float lb = p.lb_s[qidx];
        st.floor = (lb == -INFINITY) ? -INFINITY : nextafterf(lb, -INFINITY);
        st.tau = st.floor;
    }
    if (!q_valid) { st.floor = INFINITY; st.tau = INFINITY; }   // padding queries never pass the filter
    hg.counters = nullptr; hg.base = 0.0f; hg.step = 0.0f; hg.inv_step = 0.0f;
    if (hist_on && q_valid) {
        const float stp = p.hstep[qidx], lb = p.lb_s[qidx];
        if (stp > 0.0f && lb > -INFINITY) { hg.counters = p.hist + (size_t)qidx * 16; hg.base = lb; hg.step = stp; hg.inv_step = 1.0f / stp; }
    }
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.qn + (size_t)(q_valid ? qidx : 0) * MS_DIM + 64 * h);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            f32x4 v0 = src[2 * b], v1 = src[2 * b + 1];
            if (!q_valid) { v0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; v1 = v0; }
            ms_split8(v0, v1, qhi[b], qlo[b]);
        }
    }
    const float my_qlen = (mask_on && p.qlen != nullptr && q_valid) ? p.qlen[qidx] : 0.0f;
    const float qlen_eff = mask_on ? my_qlen : INFINITY;
    const float mincov_eff = mask_on ? p.mincov : 0.0f;
    float smax = -INFINITY;
    // cosine mode: scores of tile t (16 per lane: row 8 g + 4 h + j in register 4 g + j) times the length mask of their rows
    auto apply_mask = [&](f32x16 &acc, int t) __attribute__((always_inline)) {
        const f32x4 *ax = reinterpret_cast<const f32x4 *>(auxring + (t & (PF2_AUXR - 1)) * 64 + 4 * h);
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
    };
    auto visit = [&](f32x16 &sc_v, int t, bool check_rows) __attribute__((always_inline)) {
        if (mask_on) apply_mask(sc_v, t);
        const uint32_t sub_row0 = (uint32_t)(row_begin + (int64_t)t * 32) + (uint32_t)(4 * h);
        uint32_t regs = 0;                                   // registers holding a candidate of some lane (uniform)
#pragma unroll
        for (int i = 0; i < 16; ++i) regs |= (__ballot(sc_v[i] > st.tau) != 0 ? 1u : 0u) << i;
#pragma unroll 1
        while (regs != 0u) {
            const int i = __builtin_ctz(regs);
            regs &= regs - 1u;
            // register i, i uniform: a select tree of 15 v_cndmask under scalar masks (written as asm: left to itself hipcc turns
            // any such selection into a dynamic index through scratch memory)
            const uint64_t m0_ = (i & 1) ? ~0ull : 0ull, m1_ = (i & 2) ? ~0ull : 0ull, m2_ = (i & 4) ? ~0ull : 0ull, m3_ = (i & 8) ? ~0ull : 0ull;
            float l1[8], l2[4], l3[2], s;
#define MS_SEL(D, A, B, M) asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "s"(M))
#pragma unroll
            for (int j = 0; j < 8; ++j) MS_SEL(l1[j], sc_v[2 * j], sc_v[2 * j + 1], m0_);
#pragma unroll
            for (int j = 0; j < 4; ++j) MS_SEL(l2[j], l1[2 * j], l1[2 * j + 1], m1_);
            MS_SEL(l3[0], l2[0], l2[1], m2_); MS_SEL(l3[1], l2[2], l2[3], m2_);
            MS_SEL(s, l3[0], l3[1], m3_);
#undef MS_SEL
            const uint32_t row = sub_row0 + (uint32_t)(8 * (i >> 2) + (i & 3));
            bool pass = s > st.tau;
            if (check_rows) pass = pass && ((int64_t)row < row_end);
            if (pass) {
                if (hg.counters != nullptr) ms_hist_count(hg, s);
                *cand_slot(ccnt) = ms_u32x2{__float_as_uint(s), row};
                ccnt += 1;
            }
            if (__builtin_expect(__ballot(ccnt >= PF2_CAND) != 0, 0)) flush();     // some lane's buffer is full
        }
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
    f32x16 accA, accB;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA[i] = -INFINITY; accB[i] = -INFINITY; }

    auto frag_base = [&](int t) __attribute__((always_inline)) -> const f32x4 * {
        return reinterpret_cast<const f32x4 *>(smem + (size_t)(t % PF2_R) * 16384 + 16 * lane);
    };
    // STEADY: 2 <= t and t + D < ntl -- every condition of the head and the tail of a stream is known (the generic form is the
    // same code with the tests in)
    auto stage = [&](auto steady_c, int t, f32x16 &pv, f32x16 &out) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        // tile t is in registers (and the snapshot of the counters taken during the last chain): its slot is free
        PF2_T0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PF2_ACC(sp_lgkm)
        if (STEADY || t + 1 < ntl) wait_arrived(t + 1);                   // tile t + 1 has landed from every wave
        PF2_ACC(sp_sync)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = 0.0f;
        // The chain, k block by k block; behind each block's matrix instructions the two fragment reads of tile t + 1 that replace
        // its operands (past the last tile: a stale slot, never used) and ONE of everything else -- an LDS-DMA piece costs the wave
        // 8 cycles next to a matrix instruction and 60-185 in a burst.
        const f32x4 *src = frag_base(t + 1);
        const bool issuing = STEADY || t + PF2_D < ntl;         // (uniform)
#define MS_PF2_BLOCK(B)                                                                                               \
        {                                                                                                             \
            const bf16x8 fh = __builtin_bit_cast(bf16x8, fr[2 * (B)]), fl = __builtin_bit_cast(bf16x8, fr[2 * (B) + 1]); \
            out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, qhi[B], out, 0, 0, 0);                                  \
            out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, qlo[B], out, 0, 0, 0);                                  \
            out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, qhi[B], out, 0, 0, 0);                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            fr[2 * (B)] = src[64 * (2 * (B))];                                                                        \
            fr[2 * (B) + 1] = src[64 * (2 * (B) + 1)];                                                                \
        }
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
#ifdef MS_STAMP
        asm volatile("s_nop 0" : "+v"(out));
#endif
        PF2_ACC(sp_chain)
        // filter of tile t - 1: one compare per tile; the rare path only where a lane's maximum passes
        if (SAMPLE) {
            if (mask_on && t > 0) apply_mask(pv, t - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) smax = (pv[i] > smax) ? pv[i] : smax;       // (NaN scores never enter)
        } else {
            // (eight instructions; `fmaxf` compiles to ten: hipcc canonicalises the first two operands.  v_max3 ignores NaNs as fmaxf does)
            // (ONE statement: hipcc pads every asm statement with an s_nop)
            float mx;
            asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"
                "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                : "=&v"(mx) : "v"(pv[0]), "v"(pv[1]), "v"(pv[2]), "v"(pv[3]), "v"(pv[4]), "v"(pv[5]), "v"(pv[6]), "v"(pv[7]), "v"(pv[8]), "v"(pv[9]),
                  "v"(pv[10]), "v"(pv[11]), "v"(pv[12]), "v"(pv[13]), "v"(pv[14]), "v"(pv[15]));
#ifdef MS_PF2_NOVISIT
            asm volatile("" ::"v"(mx));
            if (false) {
#else
            if (__builtin_expect(__ballot(mx > st.tau) != 0 || neg_tau, 0)) {
#endif
                PF2_T0();
                if (t > 0) visit(pv, t - 1, false);
                if (mask_on) neg_tau = __ballot(st.tau < 0.0f) != 0;
#ifdef MS_STAMP
                sp_nvis += 1;
#endif
                PF2_ACC(sp_vis)
            }
        }
    };

#ifdef MS_STAMP
    const unsigned long long sp_c0 = __builtin_amdgcn_s_memtime(), sp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (ntl > 0) {
        // the first W tiles: own pieces, then (tile 0) everybody's.  With fewer than D tiles fewer pieces were issued: drain.
        if (ntl >= PF2_D) wait_own(std::integral_constant<int, PF2_D - PF2_W>{}); else ms_pf2_vmcnt<0>();
#pragma unroll
        for (int t = 0; t < PF2_W; ++t)
            if (t < ntl) publish(t);
        read_arrived(0);
        wait_arrived(0);
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
        const int fetch_phase = 2 * wave, read_phase = (2 * wave + 2) & (MS_HIST_PERIOD - 1);
        auto hist_step = [&](int t) __attribute__((always_inline)) {
            const int phase = t & (MS_HIST_PERIOD - 1);
            if (phase == fetch_phase) {
                const uint64_t hb = (uint64_t)(uintptr_t)p.hist + (uint64_t)qtile * 2048u;
                const uint32_t hb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)hb);
                const uint32_t hb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(hb >> 32));
                const uint64_t shb = ((uint64_t)hb_hi << 32) | (uint64_t)hb_lo;
                const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(ring_lds + PF2_OFF_HIST + (uint32_t)(wave & 3) * 2048u);
                ms_glds_s16_sc1<0>(dst, voff, shb);
                ms_glds_s16_sc1<1024>(dst + 1024, voff, shb);
            }
            if (phase == read_phase && t >= 2) {
                PF2_T0();
                // the two stages since then issued two tiles' pieces behind the counters' (near the end of a stream: fewer -- drain)
                if (t - 1 + PF2_D < ntl) wait_own(std::integral_constant<int, 2>{}); else ms_pf2_vmcnt<0>();
                const ms_u32x4 *hp = reinterpret_cast<const ms_u32x4 *>(smem + PF2_OFF_HIST + (wave & 3) * 2048 + r * 64);
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
                if (mask_on) neg_tau = __ballot(st.tau < 0.0f) != 0;
                PF2_ACC(sp_hist)
            }
        };
        int t = 0;
        // the body of a stream: every stage issues (t + 1 + D < ntl), two tiles per iteration ...
        for (; t + 1 + PF2_D < ntl; t += 2) {
            if (hist_on) hist_step(t);
            stage(std::true_type{}, t, accA, accB);             // accA = scores of tile t - 1 (or -inf), accB <- tile t
            stage(std::true_type{}, t + 1, accB, accA);         // accB = tile t, accA <- tile t + 1
        }
        // ... and its last D + 1 tiles, with the tests in
        for (; t + 1 < ntl; t += 2) {
            if (hist_on) hist_step(t);
            stage(std::false_type{}, t, accA, accB);
            stage(std::false_type{}, t + 1, accB, accA);
        }
        if (t < ntl) {
            stage(std::false_type{}, t, accA, accB);
            accA = accB;
        }
#ifdef MS_STAMP
        if (!SAMPLE && lane == 0 && p.stamps != nullptr && (size_t)bid * 64 + 64 <= 4 * 4 * 65536) {
            unsigned long long *o = p.stamps + ((size_t)bid * 8 + wave) * 8;
            o[0] = __builtin_amdgcn_s_memtime() - sp_c0;
            o[1] = __builtin_amdgcn_s_memrealtime() - sp_r0;
            o[2] = (unsigned long long)ntl;
            o[3] = (sp_lgkm << 32) | (sp_flow & 0xFFFFFFFFull);
            o[4] = sp_vis; o[5] = sp_nvis; o[6] = sp_chain; o[7] = (sp_hist << 32) | (sp_sync & 0xFFFFFFFFull);
        }
#endif
        // drain: the last tile's scores (accA), rows past row_end rejected
        if (SAMPLE) {
            if (mask_on) apply_mask(accA, ntl - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) smax = (accA[i] > smax) ? accA[i] : smax;
        } else {
            visit(accA, ntl - 1, true);
            flush();
        }
    }

    if (SAMPLE) {
        // the stream's entry: the two half-tile maxima of the lane pair (scores of distinct rows), larger first; rows are
        // not recorded (the bound selection reads values only), a distinct placeholder keeps the slots "occupied"
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
}

template <int KL, int NW, bool MASK>
int launch_scan_pf2(const ScanPlan &pl, const ScanParams &sp, hipStream_t st) {
    MS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_scan_pf2_kernel<KL, NW, false, MASK>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF2_LDS));
    hipLaunchKernelGGL((ms_scan_pf2_kernel<KL, NW, false, MASK>), dim3(pl.grid), dim3(64 * NW), PF2_LDS, st, sp);
    MS_LAUNCH_CHECK("ms_scan_pf2_kernel");
    return MS_OK;
}

// One non-template entry point per list length (ms_scan_pf_kl*.hip); the sample pass and the image builder live in ms_scan_pf_kl5.hip.
int ms_launch_scan_pf2_kl5(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_pf2_kl10(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_pf2_kl16(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_scan_pf2_kl32(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_sample_pf2(const ScanPlan &pl, const ScanParams &sp, hipStream_t st);
int ms_launch_pf_build_image(const float *db, int64_t n, void *image, hipStream_t st);
