// kernels_adc2.inc.hpp — included by kernels_quant.hip (namespace comet, behind adc_scan_kernel): the round-6 form of the PQ / IVFPQ asymmetric-distance
// scan for 8-bit codebooks with 4 or 8 dimensions per subspace (pq_index_search.go:243-306, ivfpq_index_search.go:285-321,350-390).
//
// What round 5's kernel (adc_scan_kernel<DSUB>) paid per item on short lists (profiles/r05_adc_ablation.txt) and what this one does about it:
//   * every item re-read the whole codebook (786 KB at M 96 x 256 x 8 floats) from L2 for its table build: 3.5 GB per launch, the L2's rate.
//     Here a workgroup takes a BATCH of up to A2_G items and walks them PHASE-MAJOR: the codewords of a phase (16 subspaces x 256 x DSUB floats = 128 KiB)
//     are held in the workgroup's REGISTERS (64 per lane) while the tables of all the batch's items are built for that phase — the slice is read
//     once per batch, the partial sums of the batch's candidates stay in registers across the phases (no parking in HBM).
//   * a wave with one chain had two LDS gathers in flight per wait: the LDS pipe ran at a third of its gather rate. Here a chain issues the 16 gathers of
//     a phase at once (one v_lshlrev_b32_sdwa per address, the table and subspace offsets in the instruction's offset field) and adds them in subspace
//     order behind counted waits.
//   * sixteen waves of 128 registers left the build nothing to keep; eight waves of 256 registers (two per SIMD: one builds while the other gathers).
// Exactness is untouched: LUT[m][k] = sum_i ((q - c)[m*DSUB+i] - cb[m][k][i])^2 with the reference's expression, order and rounding (adc_build_slab's),
// a candidate's sum = the table entries added in subspace order from 0, float32.
//
//   * a batch boundary was a chain of global round trips (ticket -> records -> rows and bounds; cursor atomic -> survivor stores), 6-14 k clocks each on the loaded
//     chip: batches are software-pipelined (adc_scan2_kernel's comment) and a batch's eight staging areas are flushed by its eight waves at once.
// Where it is used (launch_adc_scan): single-stage launches of indexes whose average list holds >= 1536 codes — configs[3]'s shape: 0.506 -> 0.455 ms; on short or very
// uneven lists and on the two-stage search's small launches round 5's kernel is faster (profiles/r06_adc_ab.txt) and stays.
//
// LDS (dynamic only — the kernel declares no static __shared__, so that the dynamic region starts at LDS address 0 and every table address is an
// immediate): [0, 32 KiB) table buffer 0, [32, 64 KiB) table buffer 1 — and, once a batch's last gathers are done, [0, 96 KiB) the batch's parked sums —
// [96, 128 KiB) the batch's residual pairs, [128, 144 KiB) eight survivor staging areas (item x query half), then a few words of state and the records of two batches.
constexpr int A2_WAVES = 8;
constexpr int A2_THREADS = A2_WAVES * 64;
constexpr int A2_CH = 6;                                        // chains (64-code blocks) per wave and item
constexpr int A2_G = 4;                                         // items per batch (even: item g's tables live in buffer g & 1)
constexpr int A2_MP = 16;                                       // subspaces per phase
constexpr int A2_SEG_CODES = A2_WAVES * A2_CH * 64;             // codes per item (3072)
constexpr int A2_RES_DIMS = 1024;                               // largest M * DSUB the kernel takes (the batch's query residuals live in LDS)
constexpr int A2_STAGE_CAP = 256;                               // survivors of one (item, query half) staged in LDS before they are appended (more: appended directly)
constexpr unsigned A2_BUF_BYTES = A2_MP * 256 * 8;              // 32 KiB: one phase of a duo's table
constexpr unsigned A2_SUMS_BYTES = A2_G * A2_CH * A2_THREADS * 8;   // the batch's final sums, parked for the epilogue loop (over the table buffers, which are free by then)
constexpr unsigned A2_RES_OFF = A2_SUMS_BYTES > 2 * A2_BUF_BYTES ? A2_SUMS_BYTES : 2 * A2_BUF_BYTES;
constexpr unsigned A2_RES_ITEM = A2_RES_DIMS * 8;               // {q_A[d] - c[d], q_B[d] - c[d]} per dimension
constexpr unsigned A2_STAGE_OFF = A2_RES_OFF + A2_G * A2_RES_ITEM;
constexpr unsigned A2_VARS_OFF = A2_STAGE_OFF + A2_G * 2 * A2_STAGE_CAP * 8;
constexpr unsigned A2_LDS_BYTES = A2_VARS_OFF + 512;
static_assert(A2_LDS_BYTES <= 160 * 1024, "adc_scan2: LDS");
static_assert(A2_RES_DIMS <= 2 * A2_THREADS, "adc_scan2: the residual pairs are formed by two rounds of the workgroup's threads");
#define A2_LDSP(T, off) ((__attribute__((address_space(3))) T*)(unsigned)(off))
// words of state behind the staging areas: the batch's ticket, per (item, half) staging counters, the batch's item records
enum { A2V_QUEUE = 0, A2V_TICKET = 1, A2V_GOT = 2, A2V_SCNT = 4, A2V_SVALID = 12, A2V_KMIN = 20, A2V_REC = 28 /*2 x A2_G x 12: the records of the batch in work and of the next one*/ };

template <int B> __device__ __forceinline__ unsigned a2_byte_x8(unsigned w) {      // (byte B of w) * 8 in ONE instruction
    unsigned r;
    if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w));
    if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w));
    if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w));
    if constexpr (B == 3) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w));
    __builtin_assume(r <= 2040u);                               // lets the table / subspace offset fold into the ds_read's immediate
    return r;
}
// one chain, one phase: 16 gathers in flight, added in subspace order
template <int BUF>
__device__ __forceinline__ void a2_chain_full(const unsigned (&w)[4], f32x2q& acc) {
    f32x2q v[16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[4 * i + 0] = *A2_LDSP(const f32x2q, a2_byte_x8<0>(w[i]) + BUF * A2_BUF_BYTES + (4 * i + 0) * 2048u);
        v[4 * i + 1] = *A2_LDSP(const f32x2q, a2_byte_x8<1>(w[i]) + BUF * A2_BUF_BYTES + (4 * i + 1) * 2048u);
        v[4 * i + 2] = *A2_LDSP(const f32x2q, a2_byte_x8<2>(w[i]) + BUF * A2_BUF_BYTES + (4 * i + 2) * 2048u);
        v[4 * i + 3] = *A2_LDSP(const f32x2q, a2_byte_x8<3>(w[i]) + BUF * A2_BUF_BYTES + (4 * i + 3) * 2048u);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) acc = acc + v[k];
    __builtin_amdgcn_sched_barrier(0);                          // one chain's sixteen gathers at a time: the scheduler would otherwise hoist every chain's (6 x 32 registers)
}

// the wave's share of a phase's codewords: subspaces 2 wid, 2 wid + 1 of the phase, four rows of 64 codewords each, the lane's codeword of every row
template <int DSUB>
__device__ __forceinline__ void a2_load_cb(const AdcArgs& a, int ph, int wid, unsigned lane, f32x4q (&cb)[2][4][DSUB / 4]) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int m = ph * A2_MP + wid * 2 + s;
        if (m >= a.M) continue;                                  // wave-uniform
        const float* __restrict__ cp = a.codebooks + ((long)m * a.Ksub + (int)lane) * DSUB;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < DSUB / 4; i++) cb[s][j][i] = *reinterpret_cast<const f32x4q*>(cp + (long)j * 64 * DSUB + i * 4);
    }
}
// One dimension of four table rows at once: acc[j] += (r2 - cb[j])^2 for the two queries of the duo (r2 = {rA, rB}; cb[j] = a register PAIR holding dimensions
// (i, i+1) of row j's codeword, HI selects which). Twelve packed operations, the four rows' chains interleaved so that no instruction follows its producer
// (hipcc serialises the rows at this register pressure: 14 s_nop and 4 v_mov per row). FIRST: the sums start here (0 + x = x).
template <bool HI, bool FIRST>
__device__ __forceinline__ void a2_dim4(f32x2q (&acc)[4], f32x2q r2, f32x2q c0, f32x2q c1, f32x2q c2, f32x2q c3) {
    f32x2q t0, t1, t2, t3;
#define A2_SUB_LO(D, C) "v_pk_add_f32 " D ", %8, " C " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define A2_SUB_HI(D, C) "v_pk_add_f32 " D ", %8, " C " op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
    if constexpr (FIRST) {
        if constexpr (!HI)
            asm(A2_SUB_LO("%4", "%9") A2_SUB_LO("%5", "%10") A2_SUB_LO("%6", "%11") A2_SUB_LO("%7", "%12")
                "v_pk_mul_f32 %0, %4, %4\n\tv_pk_mul_f32 %1, %5, %5\n\tv_pk_mul_f32 %2, %6, %6\n\tv_pk_mul_f32 %3, %7, %7"
                : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(r2), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
        else
            asm(A2_SUB_HI("%4", "%9") A2_SUB_HI("%5", "%10") A2_SUB_HI("%6", "%11") A2_SUB_HI("%7", "%12")
                "v_pk_mul_f32 %0, %4, %4\n\tv_pk_mul_f32 %1, %5, %5\n\tv_pk_mul_f32 %2, %6, %6\n\tv_pk_mul_f32 %3, %7, %7"
                : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(r2), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    } else {
        if constexpr (!HI)
            asm(A2_SUB_LO("%4", "%9") A2_SUB_LO("%5", "%10") A2_SUB_LO("%6", "%11") A2_SUB_LO("%7", "%12")
                "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(r2), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
        else
            asm(A2_SUB_HI("%4", "%9") A2_SUB_HI("%5", "%10") A2_SUB_HI("%6", "%11") A2_SUB_HI("%7", "%12")
                "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(r2), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    }
#undef A2_SUB_LO
#undef A2_SUB_HI
}
// The wave's share of one phase of one item's duo table, from the codewords in registers and the item's residual pairs in LDS (adc_build_slab's arithmetic:
// LUT[m][k] = sum_i (r[m*DSUB+i] - cb[m][k][i])^2, float32, dimension order, every operation rounded: v_pk_add_f32 with a negated operand is the IEEE
// subtraction, v_pk_mul_f32 / v_pk_add_f32 the IEEE product and sum — no fused operation).
template <int DSUB>
__device__ __forceinline__ void a2_build(int M, unsigned res_base /*LDS address of the item's residual pairs*/, int ph, int wid, unsigned lane, const f32x4q (&cb)[2][4][DSUB / 4], unsigned bufbase) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int ml = wid * 2 + s, m = ph * A2_MP + ml;
        const unsigned addr = bufbase + (unsigned)ml * 2048u + lane * 8u;
        if (m >= M) {                                            // a ragged last phase: rows of zeros (x + 0 = x: the gathers need no subspace count)
            f32x2q z; z[0] = 0.0f; z[1] = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j++) *A2_LDSP(f32x2q, addr + (unsigned)j * 512u) = z;
            continue;
        }
        f32x4q rr[DSUB / 2];                                     // {rA[i], rB[i], rA[i+1], rB[i+1]}: one address for the wave (broadcast reads)
        const unsigned ra = res_base + (unsigned)(m * DSUB) * 8u;
#pragma unroll
        for (int i = 0; i < DSUB / 2; i++) rr[i] = *A2_LDSP(const f32x4q, ra + (unsigned)i * 16u);
        f32x2q acc[4];
#pragma unroll
        for (int i = 0; i < DSUB; i += 2) {
            const f32x2q ra2 = __builtin_shufflevector(rr[i >> 1], rr[i >> 1], 0, 1), rb2 = __builtin_shufflevector(rr[i >> 1], rr[i >> 1], 2, 3);
            f32x2q c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) c[j] = (i & 2) ? __builtin_shufflevector(cb[s][j][i >> 2], cb[s][j][i >> 2], 2, 3) : __builtin_shufflevector(cb[s][j][i >> 2], cb[s][j][i >> 2], 0, 1);
            if (i == 0) a2_dim4<false, true>(acc, ra2, c[0], c[1], c[2], c[3]); else a2_dim4<false, false>(acc, ra2, c[0], c[1], c[2], c[3]);
            a2_dim4<true, false>(acc, rb2, c[0], c[1], c[2], c[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) *A2_LDSP(f32x2q, addr + (unsigned)j * 512u) = acc[j];
    }
}

#ifdef A2_TRACE
// s_memtime stamps of workgroup 8's waves 0 (early) and 4 (late), batch A2_TRACE_BATCH: per step [step start, after the first barrier / code-word requests, after the
// build slot, after the late barrier, after the gathers]; row 63 of a wave: [batch start, prologue done, loop done, sums parked, epilogues done]
__device__ unsigned long long a2_trace_buf[2 * 64 * 8];
#define A2_STAMP(STEP, SLOT) do { if (blockIdx.x == 8 && trace_on && (wid == 0 || wid == 4) && lane == 0 && (STEP) < 64) a2_trace_buf[((wid ? 64 : 0) + (STEP)) * 8 + (SLOT)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifndef A2_TRACE_BATCH
#define A2_TRACE_BATCH 2
#endif
#else
#define A2_STAMP(STEP, SLOT) do { } while (0)
#endif
// LDS-only barrier: the fences order LDS accesses only (s_waitcnt lgkmcnt(0)), so that code-word and codeword loads stay in flight across it
__device__ __forceinline__ void a2_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A batch = up to A2_G items; STEP k = (phase k / A2_G, item k % A2_G): table (k) lives in buffer k & 1 (A2_G is even), is built by all eight waves (two
// subspaces each) and gathered from by all eight. The four EARLY waves run  barrier_k ; B(k+1) ; G(k),  the four LATE waves  B(k) ; barrier_k ; G(k)  — the
// same order of events as "gather, then build the next" — so that on every SIMD one wave is in its VALU-heavy build while the other is in its LDS-bound
// gathers. The item loop is unrolled (a batch's partial sums sit in registers indexed by the item); the epilogue is one copy, fed from LDS.
// Batches are software-pipelined: a global round trip costs 6-14 k clocks on a loaded chip (s_memtime trace) and a batch boundary is a chain of them
// (ticket -> records -> query / centroid rows and bounds; cursor atomic -> survivor stores). Wave 0 takes the NEXT batch's ticket and fetches its records
// during the last phase of the batch in work; every wave requests the next batch's rows, bounds and first codeword slice before it runs the epilogue of
// the batch in work, and forms the residual pairs behind the flush.
template <int DSUB>
__global__ __launch_bounds__(A2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void adc_scan2_kernel(const AdcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float a2_lds[];  // declared so that the launch's dynamic LDS is this kernel's; addressed by absolute offsets below
    const unsigned lane = threadIdx.x & 63u, voff = lane * 4u;
    const int wid = RFL((int)(threadIdx.x >> 6));
    const bool late = wid >= 4;                                      // waves w and w + 4 share a SIMD
    // LDS-typed pointers throughout (a generic pointer made from an LDS offset is a flat address without the aperture base: a memory fault)
    __attribute__((address_space(3))) int* const vars = A2_LDSP(int, A2_VARS_OFF);
    __attribute__((address_space(3))) unsigned* const uvars = A2_LDSP(unsigned, A2_VARS_OFF);
    __attribute__((address_space(3))) unsigned long long* const stage0 = A2_LDSP(unsigned long long, A2_STAGE_OFF);
    if (threadIdx.x == 0 && (unsigned)(size_t)(__attribute__((address_space(3))) float*)a2_lds != 0u) __builtin_trap();   // the immediates assume it
    if (threadIdx.x < 2 * A2_G) { vars[A2V_SCNT + threadIdx.x] = 0; vars[A2V_SVALID + threadIdx.x] = 0x7FFFFFFF; uvars[A2V_KMIN + threadIdx.x] = 0xFFFFFFFFu; }
    const int M = a.M, M4 = a.M4;
    const int P = (M + A2_MP - 1) / A2_MP;
    const int dimp = M * DSUB;
    const int wgq = max(1, (int)(gridDim.x >> 3));                   // workgroups per queue
    // Up to A2_G consecutive tickets of a queue at once: while a queue is full a workgroup takes whole batches (the codeword slice is then read once per
    // A2_G items); as it drains the bites get smaller, so that the launch does not end on a few workgroups with four items each. A workgroup's first bite is
    // ONE item: the launch's first items run against bounds of +inf (every candidate survives) and seed them.
    auto take = [&](int& xq, int& got, bool first) -> int {        // wave 0, all lanes; blocking (the launch's first ticket, and a prefetch that met a drained queue)
        while (true) {
            bool has = false; int rem = 0;
            if (lane < 8u) { rem = a.qcount[lane] - __hip_atomic_load(&a.queues[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); has = rem > 0; }
            const unsigned m = (unsigned)__ballot(has) & 0xFFu;
            if (!m) return -1;
            const unsigned rot = ((m >> xq) | (m << (8 - xq))) & 0xFFu;
            const int pick = (xq + __builtin_ctz(rot)) & 7;
            const int want = first ? 1 : min(A2_G, max(1, __shfl(rem, pick, 64) / wgq));
            int t = 0;
            if (lane == 0u) t = atomicAdd(&a.queues[pick], want);
            t = __shfl(t, 0, 64);
            xq = pick;
            const int qc = a.qcount[pick];
            if (t < qc) { got = min(want, qc - t); return t; }
        }
    };
    // wave 0: ticket + records of a batch into LDS (vars[QUEUE / TICKET / GOT], the record area of parity `par`)
    auto publish = [&](int xq, int t, int got, int par) {
        if (t >= 0 && lane < (unsigned)(3 * got)) {
            const uint4 r = reinterpret_cast<const uint4*>(a.qitems + ((long)xq * a.qcap + t))[lane];      // records are 48 bytes: lane l holds 16-byte piece l
            __attribute__((address_space(3))) unsigned* rec = A2_LDSP(unsigned, A2_VARS_OFF + (A2V_REC + par * A2_G * 12) * 4 + lane * 16u);
            rec[0] = r.x; rec[1] = r.y; rec[2] = r.z; rec[3] = r.w;
        }
        if (lane == 0u) { vars[A2V_QUEUE] = xq; vars[A2V_TICKET] = t; vars[A2V_GOT] = got; }
    };
    int my_q = blockIdx.x & 7;
#ifdef A2_TRACE
    int trace_batch = 0;
    bool trace_on = false;
#endif
    int par = 0;                                                     // parity of the batch being taken in
    bool have_cur = false;                                           // a batch's sums are parked and wait for their epilogue
    int cgot = 0;
    if (wid == 0) { int got = 0; const int t = take(my_q, got, true); publish(my_q, t, got, 0); }
    a2_barrier();
    while (true) {
        // ---- (1) intake of the next batch: its ticket and records are in LDS; request everything its first step needs
        A2_STAMP(63, 4);                                             // (of the previous batch's trace row: its sums are parked)
        int nt0 = RFL(vars[A2V_TICKET]);
        if (nt0 == -2) {                                             // the prefetched ticket met a drained queue: take one the slow way (end of a launch)
            a2_barrier();
            if (wid == 0) { int got = 0; const int t = take(my_q, got, false); publish(my_q, t, got, par); }
            a2_barrier();
            nt0 = RFL(vars[A2V_TICKET]);
        }
        const int ngot = nt0 >= 0 ? RFL(vars[A2V_GOT]) : 0;
        int it_nch[A2_G], it_cblk[A2_G], it_live[A2_G], it_qA[A2_G], it_qB[A2_G];
        float va[A2_G][2], vb[A2_G][2], vc[A2_G][2];
        unsigned ntq[A2_G][2];
        f32x4q cb[2][4][DSUB / 4];
#pragma unroll
        for (int g = 0; g < A2_G; g++) {
            it_nch[g] = 0; it_cblk[g] = 0; it_live[g] = 0; it_qA[g] = 0; it_qB[g] = -1;
            ntq[g][0] = ntq[g][1] = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 2; r++) va[g][r] = vb[g][r] = vc[g][r] = 0.0f;
            if (g < ngot) {
                const __attribute__((address_space(3))) int* rec = A2_LDSP(const int, A2_VARS_OFF + (A2V_REC + (par * A2_G + g) * 12) * 4);   // {duo, start, seg_end, qA} {soA, qB, soB, base_lo} {base_hi, list, -, -}
                const int start = RFL(rec[1]), seg_end = RFL(rec[2]), list = RFL(rec[9]);
                const long base_blk = ((long)RFL(rec[8]) << 32) | (unsigned)RFL(rec[7]);
                it_qA[g] = RFL(rec[3]); it_qB[g] = RFL(rec[5]);
                const int nblk = (seg_end - start + 63) >> 6;
                it_nch[g] = wid < nblk ? (nblk - wid + A2_WAVES - 1) / A2_WAVES : 0;
                it_cblk[g] = (int)(base_blk + (start >> 6));
                it_live[g] = 1;
                // the item's rows (queryResidual[d] = q[d] - centroid[d]; PQ: a row of zeros) and the bounds the waves prune with
                const float* __restrict__ qa = a.Qp + (long)it_qA[g] * a.ldq;
                const float* __restrict__ qb = a.Qp + (long)(it_qB[g] >= 0 ? it_qB[g] : it_qA[g]) * a.ldq;   // a duo with a hole: the second half repeats the first (never read back)
                const float* __restrict__ cen = a.centroids + (long)(a.slist_is_list ? list : 0) * a.ldq;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int d = (int)threadIdx.x + r * A2_THREADS;
                    if (d < dimp) { va[g][r] = qa[d]; vb[g][r] = qb[d]; vc[g][r] = cen[d]; }
                }
                if (a.prune && it_nch[g] > 0) {
                    ntq[g][0] = __builtin_nontemporal_load(&a.tq[it_qA[g]]);
                    ntq[g][1] = it_qB[g] >= 0 ? __builtin_nontemporal_load(&a.tq[it_qB[g]]) : 0u;
                }
            }
        }
        if (nt0 >= 0) a2_load_cb<DSUB>(a, 0, wid, lane, cb);
        A2_STAMP(61, 0);
        // ---- (2) the epilogue of the batch in work (its sums are parked in LDS), under the requests above
        if (have_cur) {
            const int cpar = par ^ 1;
            unsigned Tall[A2_G][2];                                  // the filter's bounds of the whole batch at once (adjacent items of a queue are different queries of one list: none of them tightens another's bound)
#pragma unroll
            for (int g = 0; g < A2_G; g++) {
                Tall[g][0] = Tall[g][1] = 0xFFFFFFFFu;
                if (a.cand != nullptr && g < cgot) {
                    const __attribute__((address_space(3))) int* rec = A2_LDSP(const int, A2_VARS_OFF + (A2V_REC + (cpar * A2_G + g) * 12) * 4);
                    const int qA = RFL(rec[3]), qB = RFL(rec[5]);
                    Tall[g][0] = __builtin_nontemporal_load(&a.tq[qA]);
                    Tall[g][1] = qB >= 0 ? __builtin_nontemporal_load(&a.tq[qB]) : 0u;
                }
            }
            for (int g = 0; g < cgot; g++) {
                const __attribute__((address_space(3))) int* rec = A2_LDSP(const int, A2_VARS_OFF + (A2V_REC + (cpar * A2_G + g) * 12) * 4);
                const int c_start = RFL(rec[1]), c_end = RFL(rec[2]), c_qA = RFL(rec[3]), c_soA = RFL(rec[4]), c_qB = RFL(rec[5]), c_soB = RFL(rec[6]);
                const long c_base = ((long)RFL(rec[8]) << 32) | (unsigned)RFL(rec[7]);
                const int nblk = (c_end - c_start + 63) >> 6;
                const int c_nch = wid < nblk ? (nblk - wid + A2_WAVES - 1) / A2_WAVES : 0;
                f32x2q acc[A2_CH];
                unsigned Tpre[2];
                Tpre[0] = g == 0 ? Tall[0][0] : g == 1 ? Tall[1][0] : g == 2 ? Tall[2][0] : Tall[3][0];
                Tpre[1] = g == 0 ? Tall[0][1] : g == 1 ? Tall[1][1] : g == 2 ? Tall[2][1] : Tall[3][1];
#pragma unroll
                for (int c = 0; c < A2_CH; c++)
                    if (c < c_nch) acc[c] = *A2_LDSP(const f32x2q, (unsigned)((g * A2_CH + c) * A2_THREADS) * 8u + threadIdx.x * 8u);
                if (a.cand == nullptr) {
#pragma unroll
                    for (int c = 0; c < A2_CH; c++) {
                        const int j = c_start + ((c * A2_WAVES + wid) << 6) + (int)lane;
                        if (c < c_nch && j < c_end) {
                            const bool ok = a.elig ? (a.elig[(c_base << 6) + j] != 0) : true;
                            a.D[(long)c_qA * a.ldD + c_soA + j] = ok ? go_sqrt32q(acc[c][0]) : __uint_as_float(EXCLUDED_BITS);
                            if (c_qB >= 0) a.D[(long)c_qB * a.ldD + c_soB + j] = ok ? go_sqrt32q(acc[c][1]) : __uint_as_float(EXCLUDED_BITS);
                        }
                    }
                    continue;
                }
                // Fused top-K filter: adc_scan_kernel's, per item (see there for the argument); survivors are staged per (item, query half)
                if (c_nch > 0) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int q = h ? c_qB : c_qA, so = h ? c_soB : c_soA;
                        if (q < 0) continue;
                        const int sv = g * 2 + h;
                        const unsigned T = Tpre[h];
                        const unsigned Ts = __float_as_uint(__uint_as_float(T) * 1.0000005f);
                        unsigned keys[A2_CH];
                        unsigned lmin = 0xFFFFFFFFu;
#pragma unroll
                        for (int c = 0; c < A2_CH; c++) {
                            keys[c] = 0xFFFFFFFFu;
                            if (c >= c_nch) continue;
                            const int j = c_start + ((c * A2_WAVES + wid) << 6) + (int)lane;
                            bool ok = j < c_end;
                            if (ok && a.elig) ok = a.elig[(c_base << 6) + j] != 0;
                            if (ok) { keys[c] = __float_as_uint(acc[c][h]); lmin = min(lmin, keys[c]); }
                        }
                        __attribute__((address_space(3))) unsigned long long* const stg = stage0 + sv * A2_STAGE_CAP;
                        if (__ballot(lmin <= Ts) != 0ull) {
                            unsigned kth = 0xFFFFFFFFu;
                            if ((int)__builtin_popcountll(__ballot(lmin != 0xFFFFFFFFu)) >= a.K) {
                                kth = 0u;
#pragma unroll
                                for (int bit = 31; bit >= 0; bit--) {
                                    const unsigned tv = kth | ((1u << bit) - 1u);
                                    if ((int)__builtin_popcountll(__ballot(lmin <= tv)) < a.K) kth |= 1u << bit;
                                }
                            }
                            if (lane == 0 && kth < T) __hip_atomic_fetch_min(&uvars[A2V_KMIN + sv], kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned bnd = min(T, kth);
                            const unsigned bs = bnd >= 0x7F800000u ? 0x7F800000u : __float_as_uint(__uint_as_float(bnd) * 1.0000005f);
                            const float Td = go_sqrt32q(__uint_as_float(bnd));
#pragma unroll
                            for (int c = 0; c < A2_CH; c++) {
                                if (c >= c_nch) continue;
                                bool keep = keys[c] <= bs;
                                if (__ballot(keep) == 0ull) continue;
                                const float d = go_sqrt32q(acc[c][h]);
                                keep = keep && d <= Td && !(a.thr > 0.0f && d > a.thr);
                                const unsigned long long m = __ballot(keep);
                                if (m) {
                                    const int j = c_start + ((c * A2_WAVES + wid) << 6) + (int)lane;
                                    const int leader = __builtin_ctzll(m);
                                    const int cnt = (int)__builtin_popcountll(m);
                                    const unsigned long long comp = ((unsigned long long)adc_f2key(__float_as_uint(d)) << 32) | (unsigned)(so + j);
                                    const int rank = (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                                    int slot = 0;
                                    if ((int)lane == leader) slot = __hip_atomic_fetch_add(&vars[A2V_SCNT + sv], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    slot = RFL(__shfl(slot, leader, 64));
                                    if (slot + cnt <= A2_STAGE_CAP) { if (keep) stg[slot + rank] = comp; }
                                    else {
                                        if ((int)lane == leader) __hip_atomic_fetch_min(&vars[A2V_SVALID + sv], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        int base = 0;
                                        if ((int)lane == leader) base = atomicAdd(&a.cursor[q], cnt);
                                        base = RFL(__shfl(base, leader, 64));
                                        if (keep) __hip_atomic_store(&a.cand[(long)q * a.ldD + base + rank], comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            A2_STAMP(61, 1);
            if (a.cand != nullptr) {
                // Flush: the eight staging areas of the batch (item g, query half h) are appended by the eight waves at once — wave 2 g + h takes area (g, h): ONE
                // returning atomic on the query's cursor per area, all of them in flight together (round 5's "last wave to arrive appends" made one wave walk
                // several areas, a chain of atomic round trips the other waves then waited for at the next barrier).
                a2_barrier();
                const int sv = wid, fg = wid >> 1, fh = wid & 1;
                if (fg < cgot) {
                    const __attribute__((address_space(3))) int* rec = A2_LDSP(const int, A2_VARS_OFF + (A2V_REC + (cpar * A2_G + fg) * 12) * 4);
                    const int q = RFL(fh ? rec[5] : rec[3]);
                    const int n = min(min(RFL(vars[A2V_SCNT + sv]), RFL(vars[A2V_SVALID + sv])), A2_STAGE_CAP);
                    const unsigned kmin = (unsigned)RFL((int)uvars[A2V_KMIN + sv]);
                    if (q >= 0) {
                        __attribute__((address_space(3))) unsigned long long* const stg = stage0 + sv * A2_STAGE_CAP;
                        int app_lo = 0, app_hi = 0;
                        if (n > 0) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(&a.cursor[q], n);
                            base = RFL(__shfl(base, 0, 64));
                            // (agent-scope stores: written through the XCD's L2, so that a refining wave on another XCD can see them)
                            for (int i = (int)lane; i < n; i += 64) __hip_atomic_store(&a.cand[(long)q * a.ldD + base + i], stg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            app_lo = base; app_hi = base + n;
                        }
                        if (lane == 0 && kmin != 0xFFFFFFFFu) atomicMin(&a.tq[q], kmin);
                        // bound refinement from the survivors' row (adc_scan_kernel: the wave whose append crosses a power of two looks at the row's first entries)
                        if (a.refine && app_hi > app_lo && app_hi >= 16 && (31 - __builtin_clz((unsigned)app_hi)) != (31 - __builtin_clz((unsigned)max(app_lo, 1)))) {
                            const int nr = min(min(app_hi, ADC_REFINE_MAX), (int)min(a.ldD, (long)ADC_REFINE_MAX));
                            unsigned rk[ADC_REFINE_MAX / 64];
#pragma unroll
                            for (int i = 0; i < ADC_REFINE_MAX / 64; i++) {
                                const int e = i * 64 + (int)lane;
                                rk[i] = e < nr ? (unsigned)(__hip_atomic_load(&a.cand[(long)q * a.ldD + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) : 0xFFFFFFFFu;
                            }
                            int have = 0;
#pragma unroll
                            for (int i = 0; i < ADC_REFINE_MAX / 64; i++) have += (int)__builtin_popcountll(__ballot(rk[i] != 0xFFFFFFFFu));
                            if (have >= a.K) {
                                unsigned kk = 0u;
                                for (int bit = 31; bit >= 0; bit--) {
                                    const unsigned tv = kk | ((1u << bit) - 1u);
                                    int cntb = 0;
#pragma unroll
                                    for (int i = 0; i < ADC_REFINE_MAX / 64; i++) cntb += (int)__builtin_popcountll(__ballot(rk[i] <= tv));
                                    if (cntb < a.K) kk |= 1u << bit;
                                }
                                const float Dk = __uint_as_float((kk & 0x80000000u) ? (kk & 0x7FFFFFFFu) : ~kk);
                                const float Sb = (Dk * Dk) * 1.000001f;
                                if (lane == 0 && Sb == Sb && __float_as_uint(Sb) < 0x7F800000u) atomicMin(&a.tq[q], __float_as_uint(Sb));
                            }
                        }
                    }
                }
                if (lane == 0) { vars[A2V_SCNT + sv] = 0; vars[A2V_SVALID + sv] = 0x7FFFFFFF; uvars[A2V_KMIN + sv] = 0xFFFFFFFFu; }   // for the next batch (barriers away)
            }
            A2_STAMP(61, 2);
        }
        if (nt0 < 0) break;
#ifdef A2_TRACE
        trace_on = trace_batch == A2_TRACE_BATCH; trace_batch++;
#endif
        A2_STAMP(63, 0);
        // ---- (3) the new batch: residual pairs into LDS (all builds of the previous batch are over), bounds, the first table
        const int xq = RFL(vars[A2V_QUEUE]), t0 = nt0, got = ngot;
        (void)xq; (void)t0;
#pragma unroll
        for (int g = 0; g < A2_G; g++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int d = (int)threadIdx.x + r * A2_THREADS;
                if (it_live[g] && d < dimp) { f32x2q rr; rr[0] = va[g][r] - vc[g][r]; rr[1] = vb[g][r] - vc[g][r]; *A2_LDSP(f32x2q, A2_RES_OFF + (unsigned)g * A2_RES_ITEM + (unsigned)d * 8u) = rr; }
            }
        static_assert(A2_G == 4, "the item selects are written for four items");
        f32x2q bank[A2_G][A2_CH];
#pragma unroll
        for (int g = 0; g < A2_G; g++)
#pragma unroll
            for (int c = 0; c < A2_CH; c++) { bank[g][c][0] = 0.0f; bank[g][c][1] = 0.0f; }
        // fused filter: the bounds a wave prunes with (read once per batch: a stale bound is only looser), with the epilogue's 4 ulp of slack
        unsigned pT[A2_G][2];
        unsigned deadmask = 0u;                                      // wave-uniform: items none of whose candidates (of this wave) can pass any more
#pragma unroll
        for (int g = 0; g < A2_G; g++) {
            auto slack = [](unsigned t) { return t >= 0x7F800000u ? 0xFFFFFFFFu : __float_as_uint(__uint_as_float(t) * 1.0000005f); };
            pT[g][0] = (unsigned)RFL((int)slack(ntq[g][0]));
            pT[g][1] = (unsigned)RFL((int)slack(ntq[g][1]));
        }
        A2_STAMP(61, 3);
        a2_barrier();                                                // the residual pairs are in LDS; the previous batch's parked sums have been read
        A2_STAMP(61, 4);
        if (!late && it_live[0]) a2_build<DSUB>(M, A2_RES_OFF, 0, wid, lane, cb, 0u);     // early waves: table (0) in the prologue; late waves build it in step 0
        A2_STAMP(63, 1);
#ifdef A2_TRACE
        if (blockIdx.x == 8 && trace_on && (wid == 0 || wid == 4) && lane == 0) for (int g = 0; g < A2_G; g++) a2_trace_buf[((wid ? 64 : 0) + 62) * 8 + g] = (unsigned long long)(g == 0 ? it_nch[0] : g == 1 ? it_nch[1] : g == 2 ? it_nch[2] : it_nch[3]) | ((unsigned long long)got << 32);
#endif
        // wave 0's prefetch of the next batch: [probe, ticket, records] spread over the last phase's steps (each stage's round trip passes under a step)
        int pf_rem = 0, pf_pick = 0, pf_want = 0, pf_t = 0, pf_qc = 0;
        for (int ph = 0; ph < P; ph++) {
            const bool last_ph = ph + 1 == P;
#pragma unroll
            for (int g = 0; g < A2_G; g++) {
                const int gn = (g + 1) % A2_G;
                A2_STAMP(ph * A2_G + g, 0);
                if (!late) a2_barrier();                             // early: table (k) complete, buffer (k + 1) & 1 free
                // this step's code words: requested in front of the build, used behind it
                unsigned cw[A2_CH][4];
                const bool scan = it_nch[g] > 0 && !((deadmask >> g) & 1u);
                if (scan) {
                    const unsigned* cbase = a.codes + (long)it_cblk[g] * ((long)M4 * 64);
                    const adc_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)cbase, 0, 0x7FFFFFFF, 0x00020000);
                    const int w0 = ph * (A2_MP / 4);
#pragma unroll
                    for (int c = 0; c < A2_CH; c++)
                        if (c < it_nch[g]) {
#pragma unroll
                            for (int i = 0; i < 4; i++) cw[c][i] = adc_ldw(rs, voff, ((c * A2_WAVES + wid) * M4 + min(w0 + i, M4 - 1)) * 256);
                        }
                }
                if (last_ph && wid == 0) {
                    if (g == 0) {                                    // the queues' heads and lengths
                        pf_rem = 0;
                        if (lane < 8u) pf_rem = a.qcount[lane] - __hip_atomic_load(&a.queues[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (g == 1) {                             // the ticket (a queue that looked non-empty a step ago)
                        const unsigned m = (unsigned)__ballot(pf_rem > 0) & 0xFFu;
                        pf_t = -1; pf_want = 0; pf_qc = 0;
                        if (m) {
                            const unsigned rot = ((m >> my_q) | (m << (8 - my_q))) & 0xFFu;
                            pf_pick = (my_q + __builtin_ctz(rot)) & 7;
                            pf_want = min(A2_G, max(1, __shfl(pf_rem, pf_pick, 64) / wgq));
                            pf_qc = a.qcount[pf_pick];
                            pf_t = 0;
                            if (lane == 0u) pf_t = atomicAdd(&a.queues[pf_pick], pf_want);
                        }
                    } else if (g == 2) {                             // the records (or: no work anywhere / the queue drained meanwhile: the slow path decides)
                        int t = __shfl(pf_t, 0, 64), ngot2 = 0;
                        if (pf_want == 0) t = -2;                    // every queue looked empty a step ago: let the blocking take confirm it
                        else if (t >= pf_qc) t = -2;
                        else { ngot2 = min(pf_want, pf_qc - t); my_q = pf_pick; }
                        publish(my_q, t, ngot2, par ^ 1);
                    }
                }
                A2_STAMP(ph * A2_G + g, 1);
                {   // the build slot: table (x), x = k + 1 (early) or k (late)
                    const int phx = late ? ph : (g + 1 < A2_G ? ph : ph + 1);
                    const int gx = late ? g : gn;
                    if (phx < P) {
                        if (late ? it_live[g] : it_live[gn]) a2_build<DSUB>(M, A2_RES_OFF + (unsigned)gx * A2_RES_ITEM, phx, wid, lane, cb, (unsigned)(gx & 1) * A2_BUF_BYTES);
                        if (gx == A2_G - 1 && phx + 1 < P) a2_load_cb<DSUB>(a, phx + 1, wid, lane, cb);   // the phase's last build is behind the wave: the next slice lands under the gathers
                    }
                }
                A2_STAMP(ph * A2_G + g, 2);
                if (late) a2_barrier();                              // late: table (k) complete
                A2_STAMP(ph * A2_G + g, 3);
                if (scan) {
                    bool dead = false;
                    if (a.prune && ph > 0) {
                        bool alive = false;
#pragma unroll
                        for (int c = 0; c < A2_CH; c++)
                            if (c < it_nch[g]) alive = alive || __float_as_uint(bank[g][c][0]) <= pT[g][0] || __float_as_uint(bank[g][c][1]) <= pT[g][1];
                        dead = __ballot(alive) == 0ull;
                        if (dead) deadmask |= 1u << g;
                    }
                    if (!dead) {
#pragma unroll
                        for (int c = 0; c < A2_CH; c++) if (c < it_nch[g]) { if ((g & 1) == 0) a2_chain_full<0>(cw[c], bank[g][c]); else a2_chain_full<1>(cw[c], bank[g][c]); }
                    }
                }
                A2_STAMP(ph * A2_G + g, 4);
            }
        }
        A2_STAMP(63, 2);
        // the batch's sums go to LDS (over the table buffers: a barrier after the last gathers), so that ONE copy of the epilogue walks the items
        a2_barrier();
#pragma unroll
        for (int g = 0; g < A2_G; g++)
#pragma unroll
            for (int c = 0; c < A2_CH; c++)
                if (c < it_nch[g]) *A2_LDSP(f32x2q, (unsigned)((g * A2_CH + c) * A2_THREADS) * 8u + threadIdx.x * 8u) = bank[g][c];
        A2_STAMP(63, 3);
        have_cur = true; cgot = got; par ^= 1;
    }
}
