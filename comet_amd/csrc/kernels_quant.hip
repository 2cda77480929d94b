// kernels_quant.hip — k-means, coarse assignment, PQ encode, PQ lookup tables and the ADC scan
// for gfx950. Everything here reproduces the reference's float32 evaluation order exactly
// (clustering.go:119-272, pq_index.go:439-471, ivfpq_index_search.go:350-390) — see each kernel.
#include "kernels.hpp"

namespace comet {

__device__ __forceinline__ float go_sqrt32q(float x) { return (float)__builtin_sqrt((double)x); }

// ------------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------------
// dst[i][0..ld) = src[idx[i]][0..ld)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int ld, const int* __restrict__ idx, long k, float* __restrict__ dst) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = k * ld;
    if (i >= total) return;
    long r = i / ld; int col = (int)(i - r * ld);
    dst[i] = src[(long)idx[r] * ld + col];
}
void launch_gather_rows(Ctx* c, const float* src, int ld, const int32_t* idx, int64_t k, float* dst) {
    if (k <= 0) return;
    gather_rows_kernel<<<dim3((unsigned)ceil_div(k * ld, 256)), dim3(256), 0, c->stream>>>(src, ld, idx, k, dst);
    LAUNCH_CHECK();
}

// running arg-min over centroid blocks: D is kb x ldD with D[cc][v] = dist(v, centroid c0+cc).
// Strict '<' in ascending centroid order keeps the lowest index on ties (clustering.go:188-191, :265).
__global__ __launch_bounds__(256) void argmin_update_kernel(const float* __restrict__ D, long ldD, int kb, int c0, long n,
                                                            float* __restrict__ best, int* __restrict__ best_idx, int first) {
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    float b = first ? __builtin_inff() : best[v];
    int bi = first ? 0 : best_idx[v];
    for (int cc = 0; cc < kb; cc++) {
        float d = D[(long)cc * ldD + v];
        if (d < b) { b = d; bi = c0 + cc; }
    }
    best[v] = b; best_idx[v] = bi;
}
void launch_argmin_update(Ctx* c, const float* D, int64_t ldD, int kb, int c0, int64_t n, float* best, int32_t* best_idx, bool first) {
    if (n <= 0) return;
    ProfScope ps(c, "argmin_update");
    argmin_update_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(D, ldD, kb, c0, n, best, best_idx, first ? 1 : 0);
    LAUNCH_CHECK();
}

// assign[v] = new_idx[v]; *changed |= any difference (clustering.go:194-197)
__global__ __launch_bounds__(256) void apply_assign_kernel(const int* __restrict__ new_idx, int* __restrict__ assign, long n, int* __restrict__ changed) {
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int a = new_idx[v];
    if (assign[v] != a) { assign[v] = a; *changed = 1; }
}
void launch_apply_assign(Ctx* c, const int32_t* new_idx, int32_t* assign, int64_t n, int32_t* changed) {
    if (n <= 0) return;
    apply_assign_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(new_idx, assign, n, changed);
    LAUNCH_CHECK();
}
__global__ __launch_bounds__(256) void fill_i32_kernel(int* __restrict__ p, long n, int v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_i32(Ctx* c, int32_t* p, int64_t n, int32_t v) {
    if (n <= 0) return;
    fill_i32_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(p, n, v);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// centroid update (clustering.go:213-239): cluster sums are float32 sums over the member vectors in
// ascending vector order, so members are first bucketed by a STABLE counting sort (chunk histograms
// -> per-cluster prefix over chunks -> in-chunk ranks), then one lane per (cluster, dimension) walks
// its member list sequentially.
// ------------------------------------------------------------------------------------------------
constexpr int KM_CHUNK = 1024;

__global__ __launch_bounds__(256) void km_chunk_count_kernel(const int* __restrict__ assign, long n, int k, int* __restrict__ chunkcnt) {
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int a = assign[v];
    if (a >= 0) atomicAdd(&chunkcnt[(v / KM_CHUNK) * (long)k + a], 1);
}
// per cluster: exclusive prefix over chunks in place; counts[c] = total
__global__ __launch_bounds__(256) void km_chunk_prefix_kernel(int* __restrict__ chunkcnt, int nchunks, int k, int* __restrict__ counts) {
    int cidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (cidx >= k) return;
    int run = 0;
    for (int ch = 0; ch < nchunks; ch++) { int x = chunkcnt[(long)ch * k + cidx]; chunkcnt[(long)ch * k + cidx] = run; run += x; }
    counts[cidx] = run;
}
// offs[0..k] = exclusive prefix of counts (single workgroup)
__global__ __launch_bounds__(1024) void km_offsets_kernel(const int* __restrict__ counts, int k, int* __restrict__ offs) {
    __shared__ int part[1024];
    const int per = (k + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int mine = 0;
    for (int i = 0; i < per; i++) if (b0 + i < k) mine += counts[b0 + i];
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 1024; i++) { int x = part[i]; part[i] = run; run += x; } offs[k] = run; }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = 0; i < per; i++) if (b0 + i < k) { offs[b0 + i] = run; run += counts[b0 + i]; }
}
__global__ __launch_bounds__(KM_CHUNK) void km_place_kernel(const int* __restrict__ assign, long n, int k, const int* __restrict__ chunkpre,
                                                            const int* __restrict__ offs, int* __restrict__ members) {
    __shared__ int a[KM_CHUNK];
    const long v = (long)blockIdx.x * KM_CHUNK + threadIdx.x;
    const int mine = v < n ? assign[v] : -1;
    a[threadIdx.x] = mine;
    __syncthreads();
    if (mine < 0) return;
    int rank = 0;
    for (int u = 0; u < (int)threadIdx.x; u++) rank += (a[u] == mine);
    members[offs[mine] + chunkpre[(long)blockIdx.x * k + mine] + rank] = (int)v;
}
// one lane per (cluster, column); empty clusters keep their old centroid (clustering.go:236-238)
__global__ __launch_bounds__(256) void km_update_kernel(const float* __restrict__ V, int ld, const int* __restrict__ members,
                                                        const int* __restrict__ offs, const int* __restrict__ counts, int k,
                                                        float* __restrict__ centroids) {
    const int cidx = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ld) return;
    const int cnt = counts[cidx];
    if (cnt <= 0) return;
    const int* __restrict__ mem = members + offs[cidx];
    float sum = 0.0f;
    // the sum is one serial chain in member order, its operands are not: eight member rows are requested at once (a load per
    // iteration pays the index -> row round trip, ~0.5 us, once per member: 220 us for a 400-member cluster of a PQ subspace)
    constexpr int KU = 8;
    for (int i = 0; i < cnt; i += KU) {
        int m[KU]; float v[KU];
#pragma unroll
        for (int j = 0; j < KU; j++) m[j] = mem[min(i + j, cnt - 1)];
#pragma unroll
        for (int j = 0; j < KU; j++) v[j] = V[(long)m[j] * ld + col];
#pragma unroll
        for (int j = 0; j < KU; j++) if (i + j < cnt) sum = sum + v[j];          // clusterSums[c][dim] += v[dim]
    }
    centroids[(long)cidx * ld + col] = sum / (float)cnt;                         // sum / float32(clusterSize)
}
void launch_kmeans_update(Ctx* c, const float* V, int64_t n, int ld, const int32_t* assign, int k, float* centroids) {
    ProfScope ps(c, "kmeans_update");
    const int nchunks = (int)ceil_div(n, KM_CHUNK);
    int* chunkcnt = c->salloc<int>((size_t)nchunks * k);
    int* counts = c->salloc<int>(k);
    int* offs = c->salloc<int>(k + 1);
    int* members = c->salloc<int>(n);
    c->zero(chunkcnt, sizeof(int) * (size_t)nchunks * k);
    km_chunk_count_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(assign, n, k, chunkcnt); LAUNCH_CHECK();
    km_chunk_prefix_kernel<<<dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, c->stream>>>(chunkcnt, nchunks, k, counts); LAUNCH_CHECK();
    km_offsets_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(counts, k, offs); LAUNCH_CHECK();
    km_place_kernel<<<dim3(nchunks), dim3(KM_CHUNK), 0, c->stream>>>(assign, n, k, chunkcnt, offs, members); LAUNCH_CHECK();
    const int ub = ld <= 64 ? 64 : 256;       // PQ subspaces are 32 floats wide: one wave per cluster, not four of which three idle
    km_update_kernel<<<dim3((unsigned)ceil_div(ld, ub), k), dim3(ub), 0, c->stream>>>(V, ld, members, offs, counts, k, centroids); LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// PQ helpers
// ------------------------------------------------------------------------------------------------
// dst[v][j] = j < dsub ? src[v][col0 + j] : 0   (subspace extraction into a padded matrix)
__global__ __launch_bounds__(256) void extract_sub_kernel(const float* __restrict__ src, int ld_src, long n, int col0, int dsub,
                                                          float* __restrict__ dst, int ld_dst) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = n * ld_dst;
    if (i >= total) return;
    long v = i / ld_dst; int j = (int)(i - v * ld_dst);
    dst[i] = j < dsub ? src[v * ld_src + col0 + j] : 0.0f;
}
void launch_extract_sub(Ctx* c, const float* src, int ld_src, int64_t n, int col0, int dsub, float* dst, int ld_dst) {
    if (n <= 0) return;
    extract_sub_kernel<<<dim3((unsigned)ceil_div(n * ld_dst, 256)), dim3(256), 0, c->stream>>>(src, ld_src, n, col0, dsub, dst, ld_dst);
    LAUNCH_CHECK();
}
// R[v][j] = V[v][j] - C[assign[v]][j]   (ivfpq_index.go:216-224, :303-307); padding columns stay 0
__global__ __launch_bounds__(256) void residual_rows_kernel(const float* __restrict__ V, int ld, long n, const float* __restrict__ C,
                                                            const int* __restrict__ assign, float* __restrict__ R) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = n * ld;
    if (i >= total) return;
    long v = i / ld; int j = (int)(i - v * ld);
    R[i] = V[i] - C[(long)assign[v] * ld + j];
}
void launch_residual_rows(Ctx* c, const float* V, int ld, int64_t n, const float* C, const int32_t* assign, float* R) {
    if (n <= 0) return;
    residual_rows_kernel<<<dim3((unsigned)ceil_div(n * ld, 256)), dim3(256), 0, c->stream>>>(V, ld, n, C, assign, R);
    LAUNCH_CHECK();
}

// PQ encode (pq_index.go:439-471 == ivfpq_index.go:467-500): one lane per (vector, subspace m);
// codeword loop is wave-uniform so codebook entries arrive through scalar loads.
// codes_out: n rows of M4 = ceil(M/4) little-endian words (byte m of the row = code[m]).
template <int DSUB>
__global__ __launch_bounds__(256) void pq_encode_kernel(const float* __restrict__ R, int ld, long n, const float* __restrict__ codebooks,
                                                        int M, int Ksub, int dsub_rt, unsigned char* __restrict__ codes, int code_stride) {
    const int m = blockIdx.y;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    const float* __restrict__ sub = R + v * (long)ld + (long)m * dsub;
    float s[DSUB > 0 ? DSUB : 1];
    if constexpr (DSUB > 0) {
#pragma unroll
        for (int i = 0; i < DSUB; i++) s[i] = sub[i];
    }
    float best = __builtin_inff();
    int bi = 0;
    const float* __restrict__ cb = codebooks + (long)m * Ksub * dsub;
    for (int ks = 0; ks < Ksub; ks++) {
        float dist = 0.0f;
        if constexpr (DSUB > 0) {
#pragma unroll
            for (int i = 0; i < DSUB; i++) { float diff = s[i] - cb[(long)ks * DSUB + i]; float sq = diff * diff; dist = dist + sq; }
        } else {
            for (int i = 0; i < dsub; i++) { float diff = sub[i] - cb[(long)ks * dsub + i]; float sq = diff * diff; dist = dist + sq; }
        }
        if (dist < best) { best = dist; bi = ks; }
    }
    codes[v * (long)code_stride + m] = (unsigned char)bi;   // uint8(minIdx): truncates when Nbits > 8 like the reference
}
void launch_pq_encode(Ctx* c, const float* R, int ld, int64_t n, const float* codebooks, int M, int Ksub, int dsub,
                      uint8_t* codes, int code_stride) {
    if (n <= 0) return;
    ProfScope ps(c, "pq_encode");
    dim3 grid((unsigned)ceil_div(n, 256), M), blk(256);
#define ENC(D) pq_encode_kernel<D><<<grid, blk, 0, c->stream>>>(R, ld, n, codebooks, M, Ksub, dsub, codes, code_stride)
    switch (dsub) {
        case 1: ENC(1); break; case 2: ENC(2); break; case 4: ENC(4); break; case 8: ENC(8); break;
        case 16: ENC(16); break; case 32: ENC(32); break; default: ENC(0); break;
    }
#undef ENC
    LAUNCH_CHECK();
}

// interleave row-major codes into the scan layout: slot s (= 64-code block s/64, lane s%64) word w lives
// at dst[(s/64)*M4*64 + w*64 + s%64]; the source row of slot s is row_of_slot[s] (0xFFFFFFFF = padding).
__global__ __launch_bounds__(256) void interleave_codes_kernel(const unsigned* __restrict__ src, int M4, const unsigned* __restrict__ row_of_slot,
                                                               long nslots, unsigned* __restrict__ dst) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = nslots * M4;
    if (i >= total) return;
    long s = i / M4; int w = (int)(i - s * M4);
    unsigned r = row_of_slot[s];
    dst[(s >> 6) * M4 * 64 + (long)w * 64 + (s & 63)] = (r == 0xFFFFFFFFu) ? 0u : src[(long)r * M4 + w];
}
void launch_interleave_codes(Ctx* c, const uint32_t* src_words, int M4, const uint32_t* row_of_slot, int64_t nslots, uint32_t* dst) {
    if (nslots <= 0) return;
    interleave_codes_kernel<<<dim3((unsigned)ceil_div(nslots * M4, 256)), dim3(256), 0, c->stream>>>(src_words, M4, row_of_slot, nslots, dst);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// R(list) = an upper bound on the Euclidean norm of the decoded residuals (concatenated codewords) of a list's members: feeds the
// table-free lower bound of pq_bound_prep_kernel. cbn2[m][k] = |codeword|^2 in float64, rounded up; per list the maximum over its members of
// sqrt(sum_m cbn2[m][code_m]) with a 1e-5 margin (the float32 sum of M non-negative terms is within M 2^-24 of the real one).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_cb_norms_kernel(const float* __restrict__ codebooks, int M, int Ksub, int dsub, float* __restrict__ cbn2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Ksub) return;
    const float* p = codebooks + (long)i * dsub;
    double s = 0.0;
    for (int j = 0; j < dsub; j++) s += (double)p[j] * (double)p[j];
    cbn2[i] = (float)(s * 1.000001);
}
__global__ __launch_bounds__(256) void pq_list_rmax_kernel(const unsigned* __restrict__ codes_arr, int M4, const unsigned* __restrict__ row_of_slot,
                                                           const long* __restrict__ list_base, const int* __restrict__ list_len, const float* __restrict__ cbn2,
                                                           int M, int Ksub, float* __restrict__ rmax) {
    __shared__ float part[4];
    const int l = blockIdx.x, len = list_len[l];
    const long base = list_base[l];
    float mx = 0.0f;
    for (int j = threadIdx.x; j < len; j += 256) {
        const unsigned row = row_of_slot[base + j];
        const unsigned* cw = codes_arr + (long)row * M4;
        float s = 0.0f;
        for (int w = 0; w < M4; w++) {
            const unsigned word = cw[w];
#pragma unroll
            for (int b = 0; b < 4; b++) { const int m = w * 4 + b; if (m < M) s += cbn2[(long)m * Ksub + min((int)((word >> (8 * b)) & 0xFFu), Ksub - 1)]; }
        }
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) rmax[l] = (float)(sqrt((double)fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))) * 1.00002);
}
void launch_pq_list_rmax(Ctx* c, const float* codebooks, int M, int Ksub, int dsub, const uint32_t* codes_arr, int M4, const uint32_t* row_of_slot,
                         const int64_t* list_base, const int32_t* list_len, int nlist, float* rmax) {
    ScratchMark mark(c);
    float* cbn2 = c->salloc<float>((size_t)M * Ksub);
    pq_cb_norms_kernel<<<dim3((unsigned)ceil_div((int64_t)M * Ksub, 256)), dim3(256), 0, c->stream>>>(codebooks, M, Ksub, dsub, cbn2);
    pq_list_rmax_kernel<<<dim3((unsigned)nlist), dim3(256), 0, c->stream>>>(codes_arr, M4, row_of_slot, (const long*)list_base, list_len, cbn2, M, Ksub, rmax);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// probe bookkeeping shared by IVF and IVFPQ
// ------------------------------------------------------------------------------------------------
// seg_off[q][0..np] = prefix of list_len over the probed lists; cnts[q] = total candidates
__global__ __launch_bounds__(64) void probe_segments_kernel(const unsigned* __restrict__ probe_list, int ldp, const int* __restrict__ probe_cnt,
                                                            const int* __restrict__ list_len, int B, int np, int* __restrict__ seg_off,
                                                            int* __restrict__ cnts) {
    // one wave per query: lane p fetches the length of probe p (two dependent loads for the whole row instead of 2 * np in a chain), wave prefix sum
    const int q = blockIdx.x, lane = threadIdx.x;
    const int npq = probe_cnt ? probe_cnt[q] : np;
    int run = 0;
    for (int p0 = 0; p0 < np; p0 += 64) {
        const int p = p0 + lane;
        int len = 0;
        if (p < np && p < npq) len = list_len[probe_list[(long)q * ldp + p]];
        int inc = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (p < np) seg_off[(long)q * (np + 1) + p] = run + inc - len;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) { seg_off[(long)q * (np + 1) + np] = run; cnts[q] = run; }
}
// The exact coarse ranking selects np of nlist distances; a distance whose bits equal the selection's EXCLUDED sentinel (0xFFFFFFFF: a NaN with that payload,
// which only a query or centroid component with that very bit pattern produces) is dropped there, and the probe list would keep stale scratch beyond the
// count. A query that got fewer than np lists probes lists 0 .. np-1 instead: valid, unique, deterministic (its distances are NaN whatever it probes).
__global__ __launch_bounds__(64) void probe_complete_kernel(unsigned* __restrict__ probe_list, int np, const int* __restrict__ pcnt) {
    const int q = blockIdx.x;
    if (pcnt[q] >= np) return;
    for (int p = threadIdx.x; p < np; p += 64) probe_list[(long)q * np + p] = (unsigned)p;
}
void launch_probe_complete(Ctx* c, uint32_t* probe_list, int np, const int32_t* pcnt, int B) {
    if (B <= 0 || np <= 0) return;
    probe_complete_kernel<<<dim3((unsigned)B), dim3(64), 0, c->stream>>>(probe_list, np, pcnt);
    LAUNCH_CHECK();
}
void launch_probe_segments(Ctx* c, const uint32_t* probe_list, int ldp, const int32_t* probe_cnt, const int32_t* list_len, int B, int np,
                           int32_t* seg_off, int32_t* cnts) {
    if (B <= 0) return;
    probe_segments_kernel<<<dim3((unsigned)B), dim3(64), 0, c->stream>>>(probe_list, ldp, probe_cnt, list_len, B, np, seg_off, cnts);
    LAUNCH_CHECK();
}
__device__ __forceinline__ int find_probe(const int* __restrict__ so, int np, int pos) {
    // largest p with so[p] <= pos (so non-decreasing, so[np] = total); skips empty lists
    int lo = 0, hi = np;   // invariant: so[lo] <= pos < so[hi]
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (so[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}
// out_ids[q][i] = ids_of_slot[ list_base[L] + j ] for the selected candidate positions
__global__ __launch_bounds__(256) void finalize_probe_kernel(const unsigned* __restrict__ pos, int B, int k_cap, const unsigned* __restrict__ probe_list,
                                                             int ldp, const int* __restrict__ seg_off, int np, const long* __restrict__ list_base,
                                                             const unsigned* __restrict__ ids_of_slot, const int* __restrict__ zflag,
                                                             unsigned* __restrict__ out_ids, int* __restrict__ counts) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * k_cap;
    if (i < total) {
        int q = (int)(i / k_cap);
        unsigned p0 = pos[i];
        unsigned id = 0;
        if (p0 != 0xFFFFFFFFu) {
            const int* so = seg_off + (long)q * (np + 1);
            int p = find_probe(so, np, (int)p0);
            long slot = list_base[probe_list[(long)q * ldp + p]] + ((long)p0 - so[p]);
            id = ids_of_slot[slot];
        }
        out_ids[i] = id;
    }
    if (i < B && zflag && zflag[i]) counts[i] = -(int)COMET_ERR_ZERO_VECTOR;
}
void launch_finalize_probe(Ctx* c, const uint32_t* pos, int B, int k_cap, const uint32_t* probe_list, int ldp, const int32_t* seg_off, int np,
                           const int64_t* list_base, const uint32_t* ids_of_slot, const int32_t* zflag, uint32_t* out_ids, int32_t* counts) {
    long total = std::max<long>((long)B * k_cap, B);
    finalize_probe_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, c->stream>>>(pos, B, k_cap, probe_list, ldp, seg_off, np, (const long*)list_base, ids_of_slot, zflag, out_ids, counts);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// PQ asymmetric-distance search (pq_index_search.go:243-306, ivfpq_index_search.go:285-321,350-390), three kernels:
//
//  adc_order_kernel   sorts the (query, probed list) PAIRS of a sub-batch by list into SLOTS, every list's run padded to an
//                     even length: slots 2d and 2d+1 (a DUO) always name the same list, the second may be a hole. It also
//                     builds the work queues of the scan: per XCD, the (duo, segment) items that have codes to scan.
//  pq_lut_kernel      builds, for every pair, the M x KL table
//                       LUT[m][k] = sum_i ((q[m*dsub+i] - centroid[m*dsub+i]) - cb[m][k][i])^2
//                     (KL = min(Ksub,256): codes are uint8, entries >= 256 can never be addressed) in exact float32,
//                     dimension order, and stores it to HBM INTERLEAVED with its duo partner's: entry (m, k) of a duo is
//                     the float2 {LUT_A[m][k], LUT_B[m][k]}. A thread keeps ONE codeword in registers and walks the
//                     slots, so the codebook is read once per 32 pairs instead of once per pair.
//  adc_scan_kernel    persistent 1024-thread workgroups pull items from eight XCD-affine queues (adjacent duos share an XCD
//                     and therefore an L2). The scan is bound by LDS gathers — a wave's 64 random table reads hit the 32
//                     four-byte banks about 3.1 deep (measured, and what 32 balls in 32 bins give) — so every gather is
//                     made to serve TWO queries: one ds_read_b64 of the interleaved table returns both queries' entries
//                     for the code byte (same list, same codes), at the cost of one ds_read_b32. The duo's table
//                     (2 x 96 KiB at M=96, Ksub=256) does not fit in the 160 KiB LDS, so it streams through a two-deep
//                     ring of PHASES of `mp` subspaces (LDS-DMA of phase p+1 runs under the gathers of phase p, across
//                     items too); a lane's partial sums for its codes of the segment stay in registers across phases.
//                     Codes are stored as 64-code blocks, word-interleaved, so a wave reads 256 contiguous bytes per
//                     load. A lane owns one code of up to four blocks per pass (independent serial float32 chains), sums
//                     LUT[m][code[m]] in m order (the reference's serial sum), takes the correctly-rounded sqrt and
//                     writes both queries' distances.
// ------------------------------------------------------------------------------------------------
#ifndef ADC_THREADS_N
#define ADC_THREADS_N 1024
#endif
#ifndef ADC_BUF_KB
#define ADC_BUF_KB 64
#endif
// ADC_LOADERS_N of the workgroup's waves only stream table slabs (LDS-DMA); the others only gather. Vector-memory operations return in order per wave:
// with every wave issuing its share of a 64 KB slab, a wave's next 256-byte group of code words queued up behind the slab (~1 800 clocks per group,
// four groups per phase, s_memtime trace) — on short lists, where an item is a chain of such waits, that was most of an item's 30 us.
// Round 5: 8-bit codebooks build their tables in LDS (adc_scan_kernel<DSUB>), where a loader wave has nothing to stream — the default is 0 (every wave
// gathers and builds); the streamed form, left to the other codebook shapes, gained 15 % from -DADC_LOADERS_N=2 on clustered data (round 4).
#ifndef ADC_LOADERS_N
#define ADC_LOADERS_N 0
#endif
#ifndef ADC_WPE
#define ADC_WPE 4            // waves per SIMD the scan kernel is compiled for (4: 128 registers per lane)
#endif
constexpr int ADC_THREADS = ADC_THREADS_N;
constexpr int ADC_LOADERS = ADC_LOADERS_N;
constexpr int ADC_WAVES = ADC_THREADS / 64 - ADC_LOADERS;       // the gathering waves (chains, code blocks and the epilogue are laid out over these)
constexpr int ADC_CHAINS = 4;
constexpr int ADC_PASS_CODES = ADC_WAVES * ADC_CHAINS * 64;    // codes one workgroup scans per pass (one block per chain per wave)
constexpr int ADC_SEG_PASSES = 2;
constexpr int ADC_SEG_CODES = ADC_PASS_CODES * ADC_SEG_PASSES;  // codes per item: their partial sums live in registers across the phases
constexpr int ADC_XCD_CHUNK = 4;                               // adjacent duos kept on one XCD
#ifndef ADC_G_N
#define ADC_G_N 2
#endif
constexpr int ADC_G = ADC_G_N;                                       // code words (4 subspaces each) per software-pipelined group
constexpr int ADC_BUF_BYTES = ADC_BUF_KB * 1024;                       // one phase buffer; two of them in LDS
constexpr unsigned ADC_HOLE = 0xFFFFFFFFu;
constexpr int ADC_STAGE_CAP = 512;                                  // survivors of one (item, query half) staged in LDS before they are appended (more: appended directly)
constexpr int ADC_REFINE_MAX = 512;                                 // entries of a query's survivor row the bound refinement looks at
constexpr int LUT_PAIRS_PER_WG = 32;
constexpr int ORDER_MAX_LISTS = 36 * 1024;                     // counting-sort bins that fit in LDS
typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef float f32x2q __attribute__((ext_vector_type(2)));

template <bool HAS_CENTROID, int DSUB>
__global__ __launch_bounds__(256) void pq_lut_kernel(const float* __restrict__ Qp, int ld, const float* __restrict__ centroids,
                                                     const float* __restrict__ codebooks, int M, int Ksub, int KL, int kl_shift, int dsub,
                                                     const unsigned* __restrict__ probe_list, int ldp, int np,
                                                     const int* __restrict__ seg_off, const unsigned* __restrict__ order, int n_slots, int ppw,
                                                     float* __restrict__ lut, const int* __restrict__ used /*nullable: slots in use*/, int stream) {
    // workgroup = (256 >> kl_shift) consecutive subspaces x KL codewords, `ppw` (even) consecutive slots.
    extern __shared__ __attribute__((aligned(16))) float rs[];  // [ppw][mw * d] residual slices, then [ppw] live flags (as int)
    const int d = DSUB > 0 ? DSUB : dsub;
    const int mw = 256 >> kl_shift;                             // subspaces per workgroup
    const int m_base = blockIdx.x * mw;
    const int wcols = min(mw, M - m_base) * d;                  // residual columns this workgroup needs
    const int p0 = blockIdx.y * ppw;
    if (used && p0 >= used[0]) return;                          // behind the slots the order kernel filled (workgroup-uniform)
    const int pn = min(n_slots, p0 + ppw) - p0;
    int* live = reinterpret_cast<int*>(rs + (long)ppw * mw * d);
    // phase 1: every (slot, column) residual is formed once, all loads of the workgroup in flight together
    for (int e = threadIdx.x; e < pn * wcols; e += 256) {
        const int pl = e / wcols, col = e - pl * wcols;
        const unsigned pr = order[p0 + pl];
        bool lv = pr != ADC_HOLE;
        float r = 0.0f;
        if (lv) {
            const int q = (int)pr / np, pi = (int)pr - q * np;
            const int* so = seg_off + (long)q * (np + 1) + pi;
            lv = so[1] != so[0];                                // empty list / unused probe slot: table never read
            if (lv) {
                const float qv = Qp[(long)q * ld + m_base * d + col];
                if constexpr (HAS_CENTROID) r = qv - centroids[(long)probe_list[(long)q * ldp + pi] * ld + m_base * d + col];   // queryResidual[d] = q[d] - centroid[d]
                else r = qv;
            }
        }
        rs[(long)pl * mw * d + col] = r;
        if (col == 0) live[pl] = lv ? 1 : 0;
    }
    const int k = threadIdx.x & (KL - 1), mm = threadIdx.x >> kl_shift, m = m_base + mm;
    float cb[DSUB > 0 ? DSUB : 1];
    const float* __restrict__ cbp = codebooks + ((long)min(m, M - 1) * Ksub + k) * d;
    if constexpr (DSUB > 0) {
        if constexpr (DSUB % 4 == 0) {
#pragma unroll
            for (int i = 0; i < DSUB; i += 4) { const f32x4q v = *reinterpret_cast<const f32x4q*>(cbp + i); cb[i] = v[0]; cb[i + 1] = v[1]; cb[i + 2] = v[2]; cb[i + 3] = v[3]; }
        } else {
#pragma unroll
            for (int i = 0; i < DSUB; i++) cb[i] = cbp[i];
        }
    }
    __syncthreads();
    if (m >= M) return;
    // phase 2: a thread keeps ONE codeword in registers and walks the duos (LDS broadcast reads of the residual slices)
    auto entry = [&](int pl) -> float {
        const float* r = rs + (long)pl * mw * d + mm * d;
        float dsum = 0.0f;
        if constexpr (DSUB > 0) {
#pragma unroll
            for (int i = 0; i < DSUB; i++) { const float diff = r[i] - cb[i]; const float sq = diff * diff; dsum = dsum + sq; }
        } else {
            for (int i = 0; i < d; i++) { const float diff = r[i] - cbp[i]; const float sq = diff * diff; dsum = dsum + sq; }
        }
        return dsum;
    };
    f32x2q* __restrict__ out = reinterpret_cast<f32x2q*>(lut) + ((long)(p0 >> 1) * M + m) * KL + k;
    for (int pl = 0; pl < pn; pl += 2) {
        const bool la = live[pl] != 0, lb = (pl + 1 < pn) && live[pl + 1] != 0;
        if (!la && !lb) continue;
        f32x2q v;
        v[0] = la ? entry(pl) : 0.0f;
        v[1] = lb ? entry(pl + 1) : 0.0f;
        // written once, read once (by one workgroup's LDS-DMA): a batch's worth of tables beyond the caches is stored non-temporal (the
        // every-candidate search: 0.9 GB, +2 %); the 50 MB of the pruned search are better left to the L2 (non-temporal there: -3 %)
        if (stream) __builtin_nontemporal_store(v, &out[(long)(pl >> 1) * M * KL]); else out[(long)(pl >> 1) * M * KL] = v;
    }
}

// Row minima of the tables of the pairs behind every query's nearest list: rowmin[pair][m] = min_k LUT[m][k], every entry formed
// exactly as pq_lut_kernel forms it (same expression, same order), but nothing else is written: the two-stage search uses the
// serial float32 sum of a pair's row minima as a lower bound on every candidate of that (query, list) — float32 addition is
// monotone, so summing the minima in the scan's order can never exceed a candidate's own sum. Workgroup = the LUT kernel's
// (mw subspaces x KL codewords) over `ppw` consecutive PAIRS.
// Which pair of a query is scanned in stage 1 of the two-stage search (it seeds the query's bound), from the query's segment offsets `sor`
// (np + 1 prefix sums of its probed lists' local lengths). Default: the first probed list that holds anything HERE. `strict` (a sharded search
// whose ranks all-reduce their stage-1 bounds): probe 0 only — the query's globally nearest list, on the one rank that owns it; the other
// ranks seed nothing for the query (+inf) and receive the bound through the exchange. Without it every rank of an N-rank job scans one whole list
// per query in stage 1 — the nearest one it owns — and stage 1, the bulk of a pruned search, does not shrink with N at all.
__device__ __forceinline__ bool adc_is_first(const int* __restrict__ sor, int pi, int strict) {
    const bool nonempty = sor[pi + 1] != sor[pi];
    return nonempty && (strict ? pi == 0 : sor[pi] == sor[0]);
}
__device__ __forceinline__ bool adc_is_behind(const int* __restrict__ sor, int pi, int strict) {
    return sor[pi + 1] != sor[pi] && !adc_is_first(sor, pi, strict);
}
template <bool HAS_CENTROID, int DSUB>
__global__ __launch_bounds__(256) void pq_rowmin_kernel(const float* __restrict__ Qp, int ld, const float* __restrict__ centroids,
                                                        const float* __restrict__ codebooks, int M, int Ksub, int KL, int kl_shift, int dsub,
                                                        const unsigned* __restrict__ probe_list, int ldp, int np,
                                                        const int* __restrict__ seg_off, int n_pairs, int ppw, float* __restrict__ rowmin, int strict) {
    extern __shared__ __attribute__((aligned(16))) float rs[];  // [ppw][mw * d] residual slices, [ppw] live flags, [ppw][mw][nwv] partial minima
    const int d = DSUB > 0 ? DSUB : dsub;
    const int mw = 256 >> kl_shift, nwv = KL > 64 ? KL >> 6 : 1;
    const int m_base = blockIdx.x * mw;
    const int wcols = min(mw, M - m_base) * d;
    const int p0 = blockIdx.y * ppw, pn = min(n_pairs, p0 + ppw) - p0;
    int* live = reinterpret_cast<int*>(rs + (long)ppw * mw * d);
    float* wm = reinterpret_cast<float*>(live + ppw);
    for (int e = threadIdx.x; e < pn * wcols; e += 256) {
        const int pl = e / wcols, col = e - pl * wcols;
        const int pr = p0 + pl, q = pr / np, pi = pr - q * np;
        const int* so = seg_off + (long)q * (np + 1) + pi;
        const bool lv = adc_is_behind(so - pi, pi, strict);      // the pair scanned in stage 1 and empty lists have nothing to bound
        float r = 0.0f;
        if (lv) {
            const float qv = Qp[(long)q * ld + m_base * d + col];
            if constexpr (HAS_CENTROID) r = qv - centroids[(long)probe_list[(long)q * ldp + pi] * ld + m_base * d + col];
            else r = qv;
        }
        rs[(long)pl * mw * d + col] = r;
        if (col == 0) live[pl] = lv ? 1 : 0;
    }
    const int k = threadIdx.x & (KL - 1), mm = threadIdx.x >> kl_shift, m = m_base + mm;
    float cb[DSUB > 0 ? DSUB : 1];
    const float* __restrict__ cbp = codebooks + ((long)min(m, M - 1) * Ksub + k) * d;
    if constexpr (DSUB > 0) {
#pragma unroll
        for (int i = 0; i < DSUB; i++) cb[i] = cbp[i];
    }
    __syncthreads();
    auto entry = [&](int pl) -> float {
        const float* r = rs + (long)pl * mw * d + mm * d;
        float dsum = 0.0f;
        if constexpr (DSUB > 0) {
#pragma unroll
            for (int i = 0; i < DSUB; i++) { const float diff = r[i] - cb[i]; const float sq = diff * diff; dsum = dsum + sq; }
        } else {
            for (int i = 0; i < d; i++) { const float diff = r[i] - cbp[i]; const float sq = diff * diff; dsum = dsum + sq; }
        }
        return dsum;
    };
    const int gw = KL < 64 ? KL : 64;                            // lanes of a wave that share a subspace
    if (KL >= 64 && ppw == 32) {
        // 32 pairs at once: a reduce-scatter butterfly over the lanes — in the step with lane distance o every lane hands the half of its
        // values that its partner keeps across and keeps the other half, so 31 exchanges (instead of 32 x 6) leave lane l with the
        // minimum of pair (l & 31) over its 32-lane half; one more exchange joins the halves.
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) v[j] = (j < pn && live[j]) ? entry(j) : __builtin_inff();
        const int lane = threadIdx.x & 63;
#define RM_STEP(O) { const bool up = (lane & O) != 0; _Pragma("unroll") for (int j = 0; j < O; j++) { \
            const float send = up ? v[j] : v[j + O], keep = up ? v[j + O] : v[j]; v[j] = fminf(keep, __shfl_xor(send, O, 64)); } }
        RM_STEP(16) RM_STEP(8) RM_STEP(4) RM_STEP(2) RM_STEP(1)
#undef RM_STEP
        const float r0 = fminf(v[0], __shfl_xor(v[0], 32, 64));
        if (lane < 32) wm[((long)lane * mw + mm) * nwv + (k >> 6)] = r0;
    } else {
        for (int pl = 0; pl < pn; pl++) {
            if (!live[pl]) continue;                             // workgroup-uniform
            float v = entry(pl);
            for (int o = gw >> 1; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
            if ((k & (gw - 1)) == 0) wm[((long)pl * mw + mm) * nwv + (k >> 6)] = v;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < pn * mw; e += 256) {
        const int pl = e / mw, m2 = e - pl * mw;
        if (!live[pl] || m_base + m2 >= M) continue;
        float v = wm[((long)pl * mw + m2) * nwv];
        for (int w = 1; w < nwv; w++) v = fminf(v, wm[((long)pl * mw + m2) * nwv + w]);
        rowmin[(long)(p0 + pl) * M + m_base + m2] = v;
    }
}
// dead[pair] = 1 when no candidate of the pair's list can pass the query's bound: the serial float32 sum (the scan's order, m = 0 ..
// M-1) of the row minima is above it (with the 4 ulp of slack of the scan's own test). Stage-1 pairs and empty lists are "dead" for stage 2.
__global__ __launch_bounds__(256) void pq_lb_kernel(const float* __restrict__ rowmin, int M, int n_pairs, int np, const int* __restrict__ seg_off,
                                                    const unsigned* __restrict__ tq, unsigned char* __restrict__ dead, int* __restrict__ stats /*nullable: [pairs alive]*/, int strict) {
    extern __shared__ __attribute__((aligned(16))) float rml[];     // [64][M + 1]: the row minima of this workgroup's 64 pairs (coalesced loads; the sum itself is serial)
    const int i0 = blockIdx.x * 64, cnt = min(64, n_pairs - i0);
    for (int e = threadIdx.x; e < cnt * M; e += 256) { const int pl = e / M, m = e - pl * M; rml[pl * (M + 1) + m] = rowmin[(long)i0 * M + e]; }
    __syncthreads();
    if (threadIdx.x >= cnt) return;
    const int i = i0 + threadIdx.x;
    const int q = i / np, pi = i - q * np;
    const int* so = seg_off + (long)q * (np + 1) + pi;
    unsigned char dd = 1;
    const bool behind = adc_is_behind(so - pi, pi, strict);      // a non-empty list that stage 1 did not scan
    if (behind) {
        const float* r = rml + threadIdx.x * (M + 1);
        float lb = 0.0f;
        for (int m = 0; m < M; m++) lb = lb + r[m];
        const unsigned T = tq[q];
        const unsigned Ts = T >= 0x7F800000u ? 0xFFFFFFFFu : __float_as_uint(__uint_as_float(T) * 1.0000005f);
        dd = __float_as_uint(lb) > Ts ? 1 : 0;
    }
    dead[i] = dd;
    if (stats) { if (!dd) atomicAdd(&stats[0], 1); if (behind) atomicAdd(&stats[1], 1); }
}

// The lower bound in ONE kernel family, for 8-bit codebooks (KL = 256) and DSUB known at compile time (rounds 2-3: pq_bound_kernel, a wave per two pairs of a
// query walking the subspaces in order with entries formed exactly as the table forms them, the wave minimum through six LDS-crossbar shuffles; 0.10 ms per
// batch at 1M x 768, nprobe 32, B = 256. History of what was tried against it — group sizes 1 / 2 / 4 / 8 pairs (0.148 / 0.106 / 0.136 / 0.180 ms), four waves
// sharing the codeword slice through an LDS-DMA ring (0.194), two slices in flight (0.125), stopping the walk early (the pairs that die late then reach the
// scan) — is in DESIGN.md 3.5; the round-4 forms below replaced it.)
// ------------------------------------------------------------------------------------------------
// Round 4: the lower bound WITHOUT a walk. The round-3 kernel's time was its critical path — a pair that survives took 96 dependent steps of ~1 500 clocks
// (200 VALU instructions forming every entry as the table forms it + six waited-for LDS-crossbar shuffles per wave minimum): 45-60 us of walk behind ~15 us of
// set-up, whatever the other 8 000 pairs did (tools/bnd_trace.py measured the same shape on a cheaper-step walk, 0.077 / 0.107 ms at 1M / 10M), while all of
// the arithmetic is ~20 us of the chip. Two observations remove the chain:
//   * a lower bound does not have to be the table's value, only never above it. An entry's real value is |r|^2 + (|c|^2 - 2 r.c); w_k = fma-chain(-2 r_i, c_ki,
//     |c_k|^2) costs DSUB fused multiply-adds per codeword, two codewords per instruction (v_pk_fma_f32). Float32 against real arithmetic (u = 2^-24): the
//     table's entry e_k >= d_k (1 - 10 u) (one rounding per subtraction, square and addition of non-negative terms, DSUB = 8; 18 u for 16);
//     |w_k - (|c_k|^2 - 2 r.c_k)| <= (DSUB + 3) u (2 |c_k|^2 + |r|^2) (every fma rounds once on a partial sum bounded by |c|^2 + 2 sum |r_i c_i|; the stored norm
//     is the float64 sum rounded); |r_m|^2 is a float32 sum of squares (relative (DSUB + 1) u). All of it is below 60 u (2 max_k |c_k|^2 + |r_m|^2) = 3.6e-6 (..);
//     the term subtracts 1e-5 (2 cmax2_m + |r_m|^2) + 1e-30 and clamps at zero, so  term_m <= min_k e_k  always;
//   * the row minima of different subspaces do not depend on each other, and the sum does not have to be the serial one: any float32 sum of the (non-negative)
//     terms is within 100 u of their real sum, which is below the real sum of the table's entries, which is within 96 u of the table's serial sum — a factor
//     (1 - 2e-5) on the total covers both.
// So:
//   pq_bound_prep_kernel  one wave per pair: is the pair behind stage 1, the residual's norm against R(list) (the table-free bound); survivors go to a compact
//                         list (one atomic per 16 pairs);
//   pq_bound3_kernel      four waves per BND3_P surviving pairs (a quarter of the codewords each), LANE = SUBSPACE (64 per round, ceil(M / 64) rounds): a lane
//                         keeps -2 r_m of its pairs in registers and runs over its codewords, four per 16-byte load (table [k / 4][dimension | norm][m][k % 4]:
//                         a load instruction is 1 KB contiguous), two per v_pk_fma_f32; no LDS and no cross-lane traffic inside the loop. After a round the
//                         waves' minima meet in LDS, then per pair a DPP sum over the lanes and the test; a group leaves when none of its pairs is alive.
// Measured (same box, B = 256, nprobe 32): 1M x nlist 1024 (6 400 of 8 192 pairs survive the table-free tests) 0.100 -> 0.072 ms; 10M x nlist 4096 (~400 survive)
// 0.100 -> 0.032 ms. (A single wave per group — 64 dependent load-then-compute iterations, 9 KB in flight — took 0.13 ms at 10M: four waves quarter the chain.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_bound_tab3_kernel(const float* __restrict__ codebooks, int M, int dsub, float* __restrict__ btab3, float* __restrict__ cmax2) {
    __shared__ float part[4];
    const int m = blockIdx.x, k = threadIdx.x;                                      // Ksub == 256
    const float* p = codebooks + ((long)m * 256 + k) * dsub;
    float* o = btab3 + ((long)(k >> 2) * (dsub + 1) * M + m) * 4 + (k & 3);          // row stride M * 4 floats
    double sq = 0.0;
    for (int i = 0; i < dsub; i++) { o[(long)i * M * 4] = p[i]; sq += (double)p[i] * (double)p[i]; }
    o[(long)dsub * M * 4] = (float)sq;
    float mx = (float)(sq * 1.000001);                                               // cmax2[m] = the largest squared codeword norm of the subspace, rounded up
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((k & 63) == 0) part[k >> 6] = mx;
    __syncthreads();
    if (k == 0) cmax2[m] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}
void launch_pq_bound_tab3(Ctx* c, const float* codebooks, int M, int dsub, float* btab3, float* cmax2) {
    pq_bound_tab3_kernel<<<dim3((unsigned)M), dim3(256), 0, c->stream>>>(codebooks, M, dsub, btab3, cmax2);
    LAUNCH_CHECK();
}
#ifndef BND3_P_N
#define BND3_P_N 4
#endif
#ifndef BND3_WPE
#define BND3_WPE 2                    // waves per SIMD the register allocation aims at (218 VGPRs at four pairs; forcing three spills 38 registers: 0.073 -> 0.091 ms)
#endif
constexpr int BND3_P = BND3_P_N;      // pairs per workgroup of pq_bound3_kernel
constexpr int PREP_WAVES = 16;        // pairs per workgroup: ONE atomic per workgroup reserves its survivors' slots (an atomic per survivor — 6 400 same-address
                                      // atomics with a return value at 1M x nlist 1024 — cost 0.07 ms: they retire one per ~12 ns in the L2)
template <bool HAS_CENTROID>
__global__ __launch_bounds__(PREP_WAVES * 64) void pq_bound_prep_kernel(const float* __restrict__ Qp, int ld, const float* __restrict__ centroids, int dimc,
                                                            const unsigned* __restrict__ probe_list, int ldp, int np, const int* __restrict__ seg_off, int n_pairs,
                                                            const unsigned* __restrict__ tq, unsigned char* __restrict__ dead, int* __restrict__ stats,
                                                            const float* __restrict__ list_rmax, int strict, unsigned* __restrict__ plist, int* __restrict__ pcount) {
    __shared__ int s_keep[PREP_WAVES]; __shared__ int s_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = blockIdx.x * PREP_WAVES + wid;
    bool keep = false;
    if (i < n_pairs) {
        const int q = i / np, pi = i - q * np;
        if (!adc_is_behind(seg_off + (long)q * (np + 1), pi, strict)) { if (lane == 0) dead[i] = 1; }     // scanned in stage 1, or nothing to scan
        else {
            const unsigned T = tq[q];
            bool out = false;
            // A bound that needs no table entry: a candidate's sum is the float32 value of |r - x|^2 for its decoded residual x, and |r - x| >= |r| - |x| >=
            // |r| - R(list). Float32 against real arithmetic: every term and every addition of the non-negative terms loses at most one rounding, so the
            // computed sum is >= (1 - (d + M + 3) 2^-24) times the real one (< 1 - 1e-4 for d <= 1500); the norm below is a float32 sum in another order
            // (same relative bound) and R carries its own margin. The test is made with 1e-4 margins on every factor.
            if (list_rmax && T < 0x7F800000u && dimc <= 1400) {
                const unsigned L = probe_list[(long)q * ldp + pi];
                const float* qr = Qp + (long)q * ld; const float* cr = HAS_CENTROID ? centroids + (long)L * ld : nullptr;
                float v = 0.0f;
                if ((dimc & 3) == 0) {
                    for (int c4 = lane * 4; c4 < dimc; c4 += 256) {
                        f32x4q r = *reinterpret_cast<const f32x4q*>(qr + c4);
                        if (HAS_CENTROID) { const f32x4q cv = *reinterpret_cast<const f32x4q*>(cr + c4); r[0] = r[0] - cv[0]; r[1] = r[1] - cv[1]; r[2] = r[2] - cv[2]; r[3] = r[3] - cv[3]; }
                        v += r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
                    }
                } else {
                    for (int col = lane; col < dimc; col += 64) { const float r = HAS_CENTROID ? qr[col] - cr[col] : qr[col]; v += r * r; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const float gap = sqrtf(v * 0.9999f) * 0.9999f - list_rmax[L] * 1.0001f;
                out = gap > 0.0f && gap * gap * 0.9998f > __uint_as_float(T) * 1.0001f;
            }
            if (out) { if (lane == 0) { dead[i] = 1; if (stats) { atomicAdd(&stats[6], 1); atomicAdd(&stats[1], 1); } } }
            else keep = true;
        }
    }
    if (lane == 0) s_keep[wid] = keep ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < PREP_WAVES; w++) { const int k = s_keep[w]; s_keep[w] = n; n += k; }     // exclusive ranks
        s_base = n ? atomicAdd(pcount, n) : 0;
    }
    __syncthreads();
    if (keep && lane == 0) plist[s_base + s_keep[wid]] = (unsigned)i;
}
// sum over the wave as a wave-uniform value (DPP: quad swaps, half-row and row mirrors — every lane of a group holds the group's sum, so a mirror adds the other
// group's — then the two row broadcasts; lane 63 holds the total)
__device__ __forceinline__ float bnd_wave_sum(float a) {
    int ra;
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        "v_readlane_b32 %1, %0, 63\n"
        : "+v"(a), "=s"(ra));
    return __int_as_float(ra);
}
template <bool HAS_CENTROID, int DSUB, int P>
__device__ __forceinline__ void pq_bound3_body(const float* __restrict__ Qp, int ld, const float* __restrict__ centroids, const float* __restrict__ btab3,
                                               const float* __restrict__ cmax2, int M, const unsigned* __restrict__ probe_list, int ldp, int np,
                                               const unsigned* __restrict__ tq, unsigned char* __restrict__ dead, int* __restrict__ stats,
                                               const unsigned* __restrict__ plist, int cnt, int first) {
    // four waves per group of P pairs: wave w runs over codewords 64 w .. 64 w + 63 (16 loads of four), the minima meet in LDS once per round — a
    // single wave's 64 dependent load-then-compute iterations were the whole kernel time (9 KB in flight per wave: 45 us for a lone wave)
    __shared__ float xch[2][4][P][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int pr[P]; unsigned Ts[P]; float lbs[P];
    const float* qrow[P]; const float* crow[P];
    unsigned alive = 0u;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const bool have = first + p < cnt;                          // wave-uniform
        pr[p] = __builtin_amdgcn_readfirstlane((int)plist[have ? first + p : first]);       // wave-uniform: the pair's rows, bound and sums live in SGPRs
        const int q = pr[p] / np, pi = pr[p] - q * np;
        const unsigned T = (unsigned)__builtin_amdgcn_readfirstlane((int)tq[q]);
        Ts[p] = T >= 0x7F800000u ? 0xFFFFFFFFu : __float_as_uint(__uint_as_float(T) * 1.0000005f);
        qrow[p] = Qp + (long)q * ld;
        crow[p] = HAS_CENTROID ? centroids + (long)__builtin_amdgcn_readfirstlane((int)probe_list[(long)q * ldp + pi]) * ld : nullptr;
        lbs[p] = 0.0f;
        if (have) alive |= 1u << p;
    }
    alive = (unsigned)__builtin_amdgcn_readfirstlane((int)alive);
    const unsigned had = alive;
    int par = 0;
    for (int m0 = 0; m0 < M && alive; m0 += 64, par ^= 1) {
        const bool valid = m0 + lane < M;
        const int m = valid ? m0 + lane : M - 1;
        // this lane's subspace of every pair: -2 x residual, and what the smallest w_k is added to (|r_m|^2 minus the margin: see the head comment)
        f32x4q qv[P][DSUB / 4], cv[P][DSUB / 4];
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int i4 = 0; i4 < DSUB / 4; i4++) {
                qv[p][i4] = *reinterpret_cast<const f32x4q*>(qrow[p] + m * DSUB + i4 * 4);
                if (HAS_CENTROID) cv[p][i4] = *reinterpret_cast<const f32x4q*>(crow[p] + m * DSUB + i4 * 4);
            }
        const float cm2 = cmax2[m];
        float r2[P][DSUB], adjv[P], rm[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            float s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < DSUB; i++) {
                float r = qv[p][i >> 2][i & 3];
                if (HAS_CENTROID) r = r - cv[p][i >> 2][i & 3];
                s2 = s2 + r * r;
                r2[p][i] = -2.0f * r;
            }
            adjv[p] = s2 - (1.0e-5f * (2.0f * cm2 + s2) + 1.0e-30f);
            rm[p] = __builtin_inff();
        }
        const f32x4q* tab = reinterpret_cast<const f32x4q*>(btab3) + m;      // row (k4, i) at tab[(k4 * (DSUB + 1) + i) * M]
        auto load_cw = [&](int k4, f32x4q (&dst)[DSUB + 1]) {
            const f32x4q* t4 = tab + (long)min(wid * 16 + k4, 63) * (DSUB + 1) * M;
#pragma unroll
            for (int i = 0; i <= DSUB; i++) dst[i] = t4[(long)i * M];
        };
        auto quad = [&](const f32x4q (&cur)[DSUB + 1]) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < P; p++) {
                f32x2q a01 = {cur[DSUB][0], cur[DSUB][1]}, a23 = {cur[DSUB][2], cur[DSUB][3]};
#pragma unroll
                for (int i = 0; i < DSUB; i++) {
                    const f32x2q rr = {r2[p][i], r2[p][i]};
                    a01 = __builtin_elementwise_fma(rr, f32x2q{cur[i][0], cur[i][1]}, a01);
                    a23 = __builtin_elementwise_fma(rr, f32x2q{cur[i][2], cur[i][3]}, a23);
                }
                rm[p] = fminf(rm[p], fminf(fminf(a01[0], a01[1]), fminf(a23[0], a23[1])));
            }
        };
        f32x4q cwA[DSUB + 1], cwB[DSUB + 1];
        load_cw(0, cwA);
#pragma unroll 1
        for (int k4 = 0; k4 < 16; k4 += 2) {
            load_cw(k4 + 1, cwB);
            quad(cwA);
            load_cw(k4 + 2, cwA);
            quad(cwB);
        }
#pragma unroll
        for (int p = 0; p < P; p++) xch[par][wid][p][lane] = rm[p];
        __syncthreads();                                             // (the buffer of the round before last is free again: every wave passed this barrier since)
#pragma unroll
        for (int p = 0; p < P; p++) rm[p] = fminf(fminf(xch[par][0][p][lane], xch[par][1][p][lane]), fminf(xch[par][2][p][lane], xch[par][3][p][lane]));
        unsigned al = 0u;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float term = valid ? fmaxf(rm[p] + adjv[p], 0.0f) : 0.0f;
            lbs[p] = lbs[p] + bnd_wave_sum(term);
            al |= (__float_as_uint(lbs[p] * 0.99998f) <= Ts[p]) ? (1u << p) : 0u;
        }
        alive &= (unsigned)__builtin_amdgcn_readfirstlane((int)al);
    }
    if (wid == 0 && lane < P && ((had >> lane) & 1u)) {
        int mine = pr[0];
#pragma unroll
        for (int p = 1; p < P; p++) if (lane == p) mine = pr[p];
        const bool al = (alive >> lane) & 1u;
        dead[mine] = al ? 0 : 1;
        if (stats) { if (al) atomicAdd(&stats[0], 1); atomicAdd(&stats[1], 1); }
    }
}
template <bool HAS_CENTROID, int DSUB, int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BND3_WPE, BND3_WPE))) void pq_bound3_kernel(const float* __restrict__ Qp, int ld, const float* __restrict__ centroids, const float* __restrict__ btab3,
                                                       const float* __restrict__ cmax2, int M, const unsigned* __restrict__ probe_list, int ldp, int np,
                                                       const unsigned* __restrict__ tq, unsigned char* __restrict__ dead, int* __restrict__ stats,
                                                       const unsigned* __restrict__ plist, const int* __restrict__ pcount) {
    const int cnt = *pcount;
    const int first = (int)blockIdx.x * P;
    if (first >= cnt) return;
    pq_bound3_body<HAS_CENTROID, DSUB, P>(Qp, ld, centroids, btab3, cmax2, M, probe_list, ldp, np, tq, dead, stats, plist, cnt, first);
}

// order[] = the pair indices grouped by probed list (counting sort in LDS; pairs with nothing to scan go last; the
// order inside a list is whatever the atomics give — results do not depend on it). One workgroup, nlist + 1 bins in LDS.
__global__ __launch_bounds__(1024) void order_pairs_kernel(const unsigned* __restrict__ probe_list, int ldp, int np, const int* __restrict__ seg_off,
                                                           int n_pairs, int nlist, unsigned* __restrict__ order, unsigned* __restrict__ olist /*nullable: list of every sorted position*/) {
    extern __shared__ __attribute__((aligned(16))) int obin[];   // nlist + 1 counters
    __shared__ int part[1024];
    const int nb = nlist + 1, t = threadIdx.x;
    for (int i = t; i < nb; i += 1024) obin[i] = 0;
    __syncthreads();
    auto key_of = [&](int i) {
        const int q = i / np, pi = i - q * np;
        const int* so = seg_off + (long)q * (np + 1) + pi;
        return (so[1] == so[0]) ? nlist : (int)min(probe_list[(long)q * ldp + pi], (unsigned)(nlist - 1));
    };
    for (int i = t; i < n_pairs; i += 1024) atomicAdd(&obin[key_of(i)], 1);
    __syncthreads();
    const int per = (nb + 1023) / 1024, lo = t * per, hi = min(nb, lo + per);
    int s = 0;
    for (int i = lo; i < hi; i++) s += obin[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                                      // exclusive prefix of this thread's bins
    for (int i = lo; i < hi; i++) { const int cnt = obin[i]; obin[i] = run; run += cnt; }
    __syncthreads();
    for (int i = t; i < n_pairs; i += 1024) { const int k = key_of(i); const int pos = atomicAdd(&obin[k], 1); order[pos] = (unsigned)i; if (olist) olist[pos] = (unsigned)k; }
}
__global__ __launch_bounds__(256) void iota_kernel(unsigned* __restrict__ p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (unsigned)i;
}


struct __attribute__((aligned(16))) AdcRec { unsigned duo; int start, seg_end, qA, soA, qB, soB; unsigned base_lo, base_hi, pad0, pad1, pad2; };   // 48 bytes
// Slots and work queues of the ADC scan. One workgroup. `identity`: the list count does not fit the LDS bins — every pair
// becomes a duo of its own (slot 2i = pair i, slot 2i+1 a hole).
//   order[s]  pair of slot s (ADC_HOLE: none)          slist[s]  list of slot s (nlist: nothing to scan)
//   qitems    8 queues x qcap item RECORDS (everything the scan needs about a (duo, segment) item, so that it decodes an item with ONE
//             48-byte load instead of a chain of five dependent ones); queue x holds the chunks c = x (mod 8) of ADC_XCD_CHUNK adjacent duos
//   qcount    items per queue; queues[] (the scan's ticket counters) is zeroed here
__global__ __launch_bounds__(1024) void adc_order_kernel(const unsigned* __restrict__ probe_list, int ldp, int np, const int* __restrict__ seg_off,
                                                         int n_pairs, int nlist, int identity, const int* __restrict__ list_len, int n_slots,
                                                         unsigned* __restrict__ order, unsigned* __restrict__ slist, AdcRec* __restrict__ qitems, int qcap,
                                                         int* __restrict__ qcount, int* __restrict__ queues, unsigned* __restrict__ tq, int* __restrict__ cursor, int n_q, int lead,
                                                         const long* __restrict__ list_base, int stage, const unsigned char* __restrict__ dead, int* __restrict__ used,
                                                         int* __restrict__ gstats /*nullable: [2] += 64-code blocks the items cover, [3] += items, [4] += (query, block) pairs, [5] += searches*/, int strict,
                                                         unsigned long long* __restrict__ cand_init /*nullable*/, long ldD, int seg_codes /*codes per item: ADC_SEG_CODES or A2_SEG_CODES*/,
                                                         int seg_equal /*adc_scan2: a list longer than an item is cut into EQUAL segments (whole 64-code blocks) instead of full ones + a tail: a batch's steps are alike*/) {
    // workgroups 1.. (launched only where a search begins: stage <= 1 of a fused-filter search) fill the first ADC_REFINE_MAX entries of every query's survivor
    // row with all-ones: the scan's bound refinement reads a row while it is being appended to, and a slot not yet written must read as +inf
    if (blockIdx.x > 0) {
        if (!cand_init) return;
        const int per_row = (int)min(ldD, (long)ADC_REFINE_MAX);
        for (int qq = (int)blockIdx.x - 1; qq < n_q; qq += (int)gridDim.x - 1)
            for (int e = threadIdx.x; e < per_row; e += 1024) cand_init[(long)qq * ldD + e] = ~0ull;
        return;
    }
    // stage (two-stage fused search, see launch_adc_scan): 0 = every pair; 1 = only every query's nearest non-empty list, which seeds the
    // bounds; 2 = the other pairs that the lower-bound test (pq_lb_kernel) left alive. Pairs outside the stage take no slot at all.
    // lead (fused filter, np >= 2): every query's NEAREST list (probe 0) gets a duo of its own in the first 2 * n_q slots, so that
    // those pairs are scanned first and seed the per-query bound before the bulk of the candidates is tested against it.
    extern __shared__ __attribute__((aligned(16))) int obin[];   // nlist + 1 counters (not used by the identity order)
    if (tq && stage <= 1) for (int i = threadIdx.x; i < n_q; i += 1024) { tq[i] = 0x7F800000u; cursor[i] = 0; }   // fused filter: bound = +inf (float bits of a sum), no survivors yet
    __shared__ int wtot[16];
    const int nb = nlist + 1, t = threadIdx.x;
    auto key_of = [&](int i) {
        const int q = i / np, pi = i - q * np;
        const int* so = seg_off + (long)q * (np + 1) + pi;
        // every operand is fetched before anything is decided (no early return): a thread's keys are sixteen independent load groups in
        // flight instead of a chain of dependent round trips
        const int s0 = so[0], s1 = so[1], sr = so[-pi];
        const unsigned pl = probe_list[(long)q * ldp + pi];
        const unsigned char dd = stage == 2 ? dead[i] : (unsigned char)0;
        const bool first = s1 != s0 && (strict ? pi == 0 : s0 == sr);     // the pair stage 1 scans (adc_is_first)
        const bool out = (stage == 1 && !first) || (stage == 2 && (first || dd)) || s1 == s0;
        return out ? nlist : (int)min(pl, (unsigned)(nlist - 1));
    };
    for (int i = t; i < n_slots; i += 1024) { order[i] = ADC_HOLE; slist[i] = (unsigned)nlist; }
    __shared__ int s_used;                                          // slots actually in use (the queue builder below walks only those)
    if (t == 0) { s_used = n_slots; if (used) used[0] = n_slots; }
    const int lead0 = (lead && !identity) ? 2 * n_q : 0;            // slots of the leading region
    auto in_bulk = [&](int i) { return lead0 == 0 || (i % np) != 0; };
    if (identity) {
        __syncthreads();
        for (int i = t; i < n_pairs; i += 1024) { order[2 * i] = (unsigned)i; slist[2 * i] = (unsigned)key_of(i); }
    } else {
        for (int i = t; i < nb; i += 1024) obin[i] = 0;
        __syncthreads();
        if (lead0) for (int qq = t; qq < n_q; qq += 1024) { order[2 * qq] = (unsigned)(qq * np); slist[2 * qq] = (unsigned)key_of(qq * np); }
        // a pair's key costs two dependent loads: computed once, kept in registers for the scatter (up to KC pairs per thread)
        constexpr int KC = 16;
        int kc[KC];
#pragma unroll
        for (int j = 0; j < KC; j++) { const int i = t + j * 1024; kc[j] = (i < n_pairs && in_bulk(i)) ? key_of(i) : -1; }
#pragma unroll
        // (pairs with nothing to scan are not counted: they take no slot, and in the stages of the two-stage search they are nearly all
        // of the pairs — 8 k LDS atomics on the one counter obin[nlist] were 17 of this kernel's 25 us, s_memtime trace)
        for (int j = 0; j < KC; j++) if (kc[j] >= 0 && kc[j] < nlist) atomicAdd(&obin[kc[j]], 1);
        for (int i = t + KC * 1024; i < n_pairs; i += 1024) if (in_bulk(i)) { const int k = key_of(i); if (k < nlist) atomicAdd(&obin[k], 1); }
        __syncthreads();
        const int per = (nb + 1023) / 1024, lo = t * per, hi = min(nb, lo + per);
        int s = 0;
        for (int i = lo; i < hi; i++) s += (i < nlist) ? ((obin[i] + 1) & ~1) : obin[i];      // a list's run is padded to an even length
        // exclusive prefix over the 1024 threads: wave scan + the 16 wave totals (3 barriers instead of 20)
        const int lane_ = t & 63, w_ = t >> 6;
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane_ >= o) inc += v; }
        if (lane_ == 63) wtot[w_] = inc;
        __syncthreads();
        int wbase = 0;
        for (int j = 0; j < w_; j++) wbase += wtot[j];
        __syncthreads();                                            // wtot is reused by the queue builder below
        int run = lead0 + wbase + inc - s;                          // exclusive prefix of this thread's bins (behind the leading region)
        for (int i = lo; i < hi; i++) {
            const int cnt = obin[i]; obin[i] = run; run += (i < nlist) ? ((cnt + 1) & ~1) : cnt;
            if (i == nlist) { s_used = obin[i]; if (used) used[0] = obin[i]; }   // slots in use: everything behind holds pairs with nothing to scan
        }
        __syncthreads();
        // (pairs with nothing to scan — empty lists, pairs outside the stage — keep their slots as holes: no table is built for them)
#pragma unroll
        for (int j = 0; j < KC; j++) if (kc[j] >= 0 && kc[j] < nlist) { const int pos = atomicAdd(&obin[kc[j]], 1); order[pos] = (unsigned)(t + j * 1024); slist[pos] = (unsigned)kc[j]; }
        for (int i = t + KC * 1024; i < n_pairs; i += 1024) if (in_bulk(i)) { const int k = key_of(i); if (k < nlist) { const int pos = atomicAdd(&obin[k], 1); order[pos] = (unsigned)i; slist[pos] = (unsigned)k; } }
    }
    __syncthreads();
    // work queues: waves x and x + 8 build queue x (first and second half of its duos), two passes (count, then write)
    const int lane = t & 63, w = t >> 6, x = w & 7, h = w >> 3;
    const int n_duos = min(n_slots, (s_used + 1) & ~1) >> 1, n_chunks = (n_duos + ADC_XCD_CHUNK - 1) / ADC_XCD_CHUNK;
    const int nu = ((n_chunks - x + 7) >> 3) * ADC_XCD_CHUNK;       // duo positions of queue x
    const int umid = ((nu / 2 + 63) / 64) * 64, u0 = h ? min(umid, nu) : 0, u1 = h ? nu : min(umid, nu);
    auto segs_of = [&](int u, int& duo) -> int {
        duo = (x + 8 * (u / ADC_XCD_CHUNK)) * ADC_XCD_CHUNK + (u % ADC_XCD_CHUNK);
        if (u >= u1 || duo >= n_duos) return 0;
        const unsigned L = slist[2 * duo];
        return L < (unsigned)nlist ? (list_len[L] + seg_codes - 1) / seg_codes : 0;
    };
    int tot = 0;
    for (int u = u0 + lane; u < u1; u += 64) { int duo; tot += segs_of(u, duo); }
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) wtot[w] = tot;
    __syncthreads();
    int base = h ? wtot[x] : 0;
    int st_blocks = 0, st_items = 0, st_qblocks = 0;                // what the scan will move / score (bench roofline)
    constexpr int NBK = 4;                                          // 64-duo blocks gathered together: their (dependent) loads overlap
    for (int ub0 = u0; ub0 < u1; ub0 += 64 * NBK) {
        AdcRec r[NBK]; int ns[NBK], len[NBK];
#pragma unroll
        for (int b = 0; b < NBK; b++) {
            int duo; ns[b] = segs_of(ub0 + b * 64 + lane, duo);
            r[b].duo = (unsigned)duo; len[b] = 0;
            if (ns[b] > 0) {
                const unsigned pa = order[2 * duo], pb = order[2 * duo + 1], L = slist[2 * duo];
                len[b] = list_len[L];
                const long bb = list_base[L] >> 6;                  // list bases are multiples of 64
                r[b].qA = (int)pa / np; r[b].soA = seg_off[(long)r[b].qA * (np + 1) + ((int)pa - r[b].qA * np)];
                r[b].qB = -1; r[b].soB = 0;
                if (pb != ADC_HOLE) { r[b].qB = (int)pb / np; r[b].soB = seg_off[(long)r[b].qB * (np + 1) + ((int)pb - r[b].qB * np)]; }
                r[b].base_lo = (unsigned)(bb & 0xFFFFFFFFl); r[b].base_hi = (unsigned)(bb >> 32); r[b].pad0 = L /*adc_scan2: the list (its centroid row)*/; r[b].pad1 = r[b].pad2 = 0;
                const int nblk = (len[b] + 63) >> 6;
                st_blocks += nblk; st_items += ns[b]; st_qblocks += nblk * (r[b].qB >= 0 ? 2 : 1);
            }
        }
#pragma unroll
        for (int b = 0; b < NBK; b++) {
            int inc = ns[b];                                        // inclusive wave scan
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
            const int off = base + inc - ns[b];
            for (int sgi = 0; sgi < ns[b]; sgi++) if (off + sgi < qcap) {
                const int segw = seg_equal ? (((len[b] + ns[b] - 1) / ns[b] + 63) & ~63) : seg_codes;     // (<= seg_codes: seg_codes is a multiple of 64 and ns = ceil(len / seg_codes))
                r[b].start = min(sgi * segw, (len[b] + 63) & ~63); r[b].seg_end = min(len[b], r[b].start + segw);     // (a list of > 48 items' worth of codes: the rounding can leave a last, empty segment)
                qitems[(long)x * qcap + off + sgi] = r[b];
            }
            base += __shfl(inc, 63);
        }
    }
    if (h && lane == 0) { qcount[x] = min(wtot[x] + wtot[x + 8], qcap); queues[x] = 0; }
    if (gstats) {
        for (int o = 32; o > 0; o >>= 1) { st_blocks += __shfl_xor(st_blocks, o); st_items += __shfl_xor(st_items, o); st_qblocks += __shfl_xor(st_qblocks, o); }
        if (lane == 0 && st_items) { atomicAdd(&gstats[2], st_blocks); atomicAdd(&gstats[3], st_items); atomicAdd(&gstats[4], st_qblocks); }
        if (t == 0 && stage <= 1) atomicAdd(&gstats[5], 1);
    }
}

// One wave, C blocks of a pass (chains), one PHASE of the duo's table: subspaces [4*w_lo, m_hi), table rows relative to 4*w_lo.
// Code words come through buffer loads: a descriptor per item (base = the segment's first block), the wave-uniform part of
// the address in soffset, the lane in voffset — no 64-bit per-lane addresses in VGPRs. `cur` holds the code words of the
// first group; the main loop runs over groups of ADC_G full words (4 codes each) without a branch so that the next group's
// loads and this group's gathers overlap the add chains; the last iteration fetches the first group of the wave's NEXT call
// (descriptor nrs, word nx_word of the block at byte offset nx_off; blocks past the segment read other lists' codes or the
// pad behind the last list — readable, never used). Words past the last full group are handled one code at a time.
#define RFL(x) __builtin_amdgcn_readfirstlane(x)
typedef __amdgpu_buffer_rsrc_t adc_rsrc_t;
struct AdcItem {
    int live, start, seg_end;
    int qA, soA, qB, soB;       // the two queries and the offsets of this list inside their candidate rows (qB < 0: no second query)
    long base_blk, duo;
};
__device__ __forceinline__ unsigned adc_ldw(adc_rsrc_t r, unsigned voff, int soff) { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0); }
template <int C>
__device__ __forceinline__ void adc_chains2(const f32x2q* __restrict__ lut, int w_lo, int m_hi, int M4, int KL, adc_rsrc_t rs, int off, int stride16b,
                                            adc_rsrc_t nrs, int nx_off, int nx_word, unsigned voff, unsigned (&cur)[ADC_CHAINS][ADC_G], f32x2q (&acc)[ADC_CHAINS]) {
    constexpr int G = ADC_G;
    const int ngroups = ((m_hi >> 2) - w_lo) / G;
    unsigned nxt[ADC_CHAINS][G];
    int nxw[G];                                      // byte offsets of the next call's first group (clamped: a table of < G words)
#pragma unroll
    for (int i = 0; i < G; i++) nxw[i] = nx_off + min(nx_word + i, M4 - 1) * 256;
    for (int g = 0; g < ngroups; g++) {
        if (g + 1 < ngroups) {
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int i = 0; i < G; i++) nxt[c][i] = adc_ldw(rs, voff, off + c * stride16b + (w_lo + (g + 1) * G + i) * 256);
#pragma unroll
            for (int c = C; c < ADC_CHAINS; c++)
#pragma unroll
                for (int i = 0; i < G; i++) nxt[c][i] = cur[c][i];
        } else {
#pragma unroll
            for (int c = 0; c < ADC_CHAINS; c++)
#pragma unroll
                for (int i = 0; i < G; i++) nxt[c][i] = adc_ldw(nrs, voff, nxw[i] + c * stride16b);
        }
        const f32x2q* l0 = lut + (long)g * (G * 4) * KL;
        if constexpr (C <= 2) {
            // one or two chains (lists of a few hundred codes: the short-list regime): a whole group's gathers of a chain at once — 8 (C = 1: both words) or
            // 2 x 4 (C = 2: word by word) in flight per wait instead of 2 C. With two in flight the LDS pipe ran at a third of its gather rate (round 5's ablation).
            constexpr int WI = C == 1 ? G : 1;                                   // words per round
#pragma unroll
            for (int i0 = 0; i0 < G; i0 += WI) {
                f32x2q v[C][WI * 4];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int i = 0; i < WI; i++)
#pragma unroll
                        for (int b = 0; b < 4; b++) v[c][i * 4 + b] = l0[((i0 + i) * 4 + b) * KL + ((cur[c][i0 + i] >> (8 * b)) & 255u)];
#pragma unroll
                for (int e = 0; e < WI * 4; e++)
#pragma unroll
                    for (int c = 0; c < C; c++) { acc[c][0] = acc[c][0] + v[c][e][0]; acc[c][1] = acc[c][1] + v[c][e][1]; }
            }
        } else {
#pragma unroll
        for (int i = 0; i < G; i++) {
#pragma unroll
            for (int hb = 0; hb < 4; hb += 2) {          // two code bytes of every chain at a time: 2*C gathers in flight per wave
                f32x2q v[C][2];
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const unsigned w = cur[c][i] >> (8 * hb);
                    v[c][0] = l0[(i * 4 + hb) * KL + (w & 255u)]; v[c][1] = l0[(i * 4 + hb + 1) * KL + ((w >> 8) & 255u)];
                }
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int c = 0; c < C; c++) { acc[c][0] = acc[c][0] + v[c][b][0]; acc[c][1] = acc[c][1] + v[c][b][1]; }
            }
        }
        }
#pragma unroll
        for (int c = 0; c < ADC_CHAINS; c++)
#pragma unroll
            for (int i = 0; i < G; i++) cur[c][i] = nxt[c][i];
    }
    const int w_end = (m_hi + 3) >> 2;
    if (w_lo + ngroups * G < w_end) {                 // tail words (M not a multiple of 4*G inside this phase)
        for (int w0 = w_lo + ngroups * G; w0 < w_end; w0++) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                const unsigned w = adc_ldw(rs, voff, off + c * stride16b + w0 * 256);
                for (int bb = 0; bb < 4 && w0 * 4 + bb < m_hi; bb++) {
                    const f32x2q v = lut[(long)((w0 - w_lo) * 4 + bb) * KL + ((w >> (8 * bb)) & 255u)];
                    acc[c][0] = acc[c][0] + v[0]; acc[c][1] = acc[c][1] + v[1];
                }
            }
        }
    }
    if (ngroups == 0) {                                // the first group was never consumed (a phase of < G words, or the zero-width phase of a pruned wave): `cur` must still become the next call's
#pragma unroll
        for (int c = 0; c < ADC_CHAINS; c++)
#pragma unroll
            for (int i = 0; i < G; i++) cur[c][i] = adc_ldw(nrs, voff, nxw[i] + c * stride16b);
    }
}

struct AdcArgs {
    const float* lutg; const unsigned* codes; const long* list_base; const int* list_len; const int* seg_off; const unsigned char* elig;
    const unsigned* order; const unsigned* slist; const AdcRec* qitems; const int* qcount; int* queues; float* D;
    long ldD; int M, KL, mp, M4, np, qcap;
    // fused top-K filter (cand != nullptr; K in [1, 64]): no distance matrix — survivors of the per-query running bound tq[] go to
    // cand[q * ldD + cursor[q]++] as (order-preserving key << 32 | position in the query's candidate row)
    unsigned long long* cand; int* cursor; unsigned* tq; int K; float thr;
    int prune;                  // fused filter: waves skip the remaining phases of an item once all their partial sums exceed the bounds
    int refine;                 // fused filter: the bound is refined from the survivors' row (the caller filled the first ADC_REFINE_MAX entries of every row with ~0)
    // tables built in LDS by the scanning workgroup (adc_scan_kernel<DSUB>): the queries, the coarse centroids (null: PQ, residual = query) and the codebooks
    const float* Qp; const float* centroids /*PQ: a row of zeros*/; const float* codebooks; int ldq, Ksub, slist_is_list /*0: PQ, one list, the centroid row is row 0*/;
};
// The table slabs' LDS-DMA pieces are issued from inline asm (16 bytes per lane from sbase + voff to LDS byte address lds_addr + lane * 16,
// non-temporal): with the builtin, hipcc's wait-count pass sees a pending access that may complete on either counter and turns EVERY wait of the
// item — each group's code words, each batch of table gathers — into vmcnt(0) / lgkmcnt(0) (119 + 138 of them in the round-3 binary): no gather stayed
// in flight under the add chains, and every code-word wait drained the slab as well. The kernel waits for a slab itself (ADC_TABLE_WAIT) in front
// of the phase barrier. M0 is saved and restored inside the statement.
__device__ __forceinline__ void adc_dma16(const void* sbase /*wave-uniform*/, unsigned voff, unsigned lds_addr /*wave-uniform*/) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)), la = __builtin_amdgcn_readfirstlane(lds_addr);
    const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(bu), "s"(la) : "memory");
}
#define ADC_TABLE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ unsigned adc_f2key(unsigned u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
#ifdef ADC_TRACE
// s_memtime stamps of workgroup 8's thread 0 (build with -DADC_TRACE; tools/adc_trace.py reads them): per item [start, then per phase: after the barrier,
// after the gathers ..., epilogue done]
__device__ unsigned long long adc_trace_buf[32 * 16];
// (thread 0 = wave 0, which builds its share of the next slab AFTER its gathers, fills rows 0..15; thread 64 = wave 1, which builds BEFORE them, rows 16..31)
#define ADC_STAMP(ITEM, SLOT) do { if (blockIdx.x == 8 && (threadIdx.x == 0 || threadIdx.x == 64) && (ITEM) < 16) adc_trace_buf[((ITEM) + (threadIdx.x ? 16 : 0)) * 16 + (SLOT)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ADC_STAMP(ITEM, SLOT) do { } while (0)
#endif
// One phase of a duo's table built straight into its LDS buffer by the workgroup that will gather from it (KL = 256): entry (m, k) = {LUT_A[m][k], LUT_B[m][k]},
// LUT[m][k] = sum_i ((q[m*DSUB+i] - centroid[m*DSUB+i]) - cb[m][k][i])^2 — the expression, the order and the rounding of pq_lut_kernel (and of
// ivfpq_index_search.go:350-375), the two queries of the duo in the two halves of packed float32 operations. The streamed form writes every table to HBM
// (pq_lut_kernel: 0.8 GB per batch at M = 96) and reads it back through LDS-DMA; on lists of a few thousand codes an item is then a chain of slab waits
// (DESIGN.md 3.5). Here a slab never leaves the CU: every wave forms 1/16 of the next phase's entries (4 units of 128 codewords: 3 packed VALU
// operations per dimension and codeword for the two queries together) before it gathers from the current one; what it reads is the codebook slice of
// the phase (from L2) and 2 x DSUB floats of queries / centroid per subspace.
template <int DSUB>
__device__ __forceinline__ void adc_build_slab(const AdcArgs& a, int live, long duo, int qA, int qB, int ph, f32x2q* __restrict__ slab, int wid, unsigned lane) {
    if (!live) return;
    const int m0 = ph * a.mp, nm = min(a.M, m0 + a.mp) - m0;          // the subspaces of this phase
    const float* __restrict__ qa = a.Qp + (long)qA * a.ldq;
    const float* __restrict__ qb = a.Qp + (long)(qB >= 0 ? qB : qA) * a.ldq;   // a duo with a hole: the second half repeats the first (never read back)
    // the list's centroid (PQ: a row of zeros — x - 0 is x, bit for bit): the address is formed without waiting for the list id where it can be
    const float* __restrict__ cen = a.centroids + (long)(a.slist_is_list ? RFL((int)a.slist[2 * duo]) : 0) * a.ldq;
    constexpr int NW = ADC_THREADS / 64;
    const int per = (nm + NW - 1) / NW;                               // whole subspaces per wave (32 per phase: 2)
    const int ml_hi = min(nm, (wid + 1) * per);
    for (int ml = wid * per; ml < ml_hi; ml++) {
        const int m = m0 + ml;
        // every load of the subspace in flight at once: 3 x DSUB uniform floats (queries, centroid) and the lane's four codewords
        f32x4q va[DSUB / 4], vb[DSUB / 4], vc[DSUB / 4], cw[4][DSUB / 4];
#pragma unroll
        for (int i = 0; i < DSUB / 4; i++) {
            va[i] = *reinterpret_cast<const f32x4q*>(qa + m * DSUB + i * 4); vb[i] = *reinterpret_cast<const f32x4q*>(qb + m * DSUB + i * 4);
            vc[i] = *reinterpret_cast<const f32x4q*>(cen + m * DSUB + i * 4);
        }
        const float* __restrict__ cp = a.codebooks + ((long)m * a.Ksub + (int)lane) * DSUB;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < DSUB / 4; i++) cw[j][i] = *reinterpret_cast<const f32x4q*>(cp + (long)j * 64 * DSUB + i * 4);
        f32x2q r2[DSUB];
#pragma unroll
        for (int i = 0; i < DSUB; i++) { r2[i][0] = va[i >> 2][i & 3] - vc[i >> 2][i & 3]; r2[i][1] = vb[i >> 2][i & 3] - vc[i >> 2][i & 3]; }   // queryResidual[d] = q[d] - centroid[d]
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x2q acc; acc[0] = 0.0f; acc[1] = 0.0f;
#pragma unroll
            for (int i = 0; i < DSUB; i++) {
                f32x2q c2; c2[0] = cw[j][i >> 2][i & 3]; c2[1] = c2[0];
                const f32x2q diff = r2[i] - c2;
                const f32x2q sq = diff * diff;
                acc = acc + sq;
            }
            slab[ml * 256 + j * 64 + (int)lane] = acc;
        }
    }
}

// BUILD = 0: the duo tables are streamed from HBM (built by pq_lut_kernel); BUILD = DSUB (4 / 8 / 16; KL = 256): built in LDS by this workgroup
template <int BUILD>
__global__ __launch_bounds__(ADC_THREADS) __attribute__((amdgpu_waves_per_eu(ADC_WPE, ADC_WPE))) void adc_scan_kernel(const AdcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // two phase buffers of ADC_BUF_BYTES
    __shared__ int s_ticket[2][2];                                  // [item parity][0] queue, [1] ticket (-1: all queues drained)
    // Fused filter: an item's survivors are STAGED here per query half and flushed by the last wave of the item to arrive — one returning global atomic per
    // (item, half) on the query's cursor instead of one per wave and chain. With few queries in a batch (or loose bounds: the first items of every launch) all
    // workgroups append to the same few rows and the same-address returning atomics serialise in the L2: the epilogue was 0.52 of a 0.61 ms launch at B = 32 and
    // 0.10 of 0.52 ms at B = 256 on uniform rows (profiles/r05_adc_ablation.txt).
    __shared__ unsigned long long s_stage[2][ADC_STAGE_CAP];
    __shared__ int s_scnt[2], s_sdone[2], s_svalid[2];
    __shared__ unsigned s_kmin[2];                                  // the smallest bound the item's waves offer for the half: ONE global atomicMin per (item, half), by the flushing wave
    if (threadIdx.x < 2) { s_scnt[threadIdx.x] = 0; s_sdone[threadIdx.x] = 0; s_svalid[threadIdx.x] = 0x7FFFFFFF; s_kmin[threadIdx.x] = 0xFFFFFFFFu; }
    const unsigned lane = threadIdx.x & 63u, voff = lane * 4u;
    const int wid = RFL((int)(threadIdx.x >> 6));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const int M = a.M, KL = a.KL, mp = a.mp, M4 = a.M4;
    const long n_ent = (long)M * KL;                                // float2 entries of a duo's table
    const int P = (M + mp - 1) / mp;
    const int stride16b = ADC_WAVES * M4 * 256;                     // bytes between a wave's chains (16 blocks)
    // Eight queues, one per XCD (workgroup b runs on XCD b % 8); a drained workgroup steals from the next queue.
    // (wave 0, all lanes) lanes 0..7 look at the eight queues at once, so that a drained launch costs one round trip instead of eight
    // dependent atomics; the atomic goes to the first queue with work at or behind the workgroup's own.
    auto take = [&](int& xq) -> int {
        while (true) {
            bool has = false;
            if (lane < 8u) has = __hip_atomic_load(&a.queues[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.qcount[lane];
            const unsigned m = (unsigned)__ballot(has) & 0xFFu;
            if (!m) return -1;
            const unsigned rot = ((m >> xq) | (m << (8 - xq))) & 0xFFu;
            const int pick = (xq + __builtin_ctz(rot)) & 7;
            int t = 0;
            if (lane == 0u) t = atomicAdd(&a.queues[pick], 1);
            t = __shfl(t, 0, 64);
            xq = pick;
            if (t < a.qcount[pick]) return t;
        }
    };
    auto decode = [&](int xq, int ticket) -> AdcItem {
        AdcItem it; it.live = 0; it.start = it.seg_end = 0; it.qA = it.soA = it.soB = 0; it.qB = -1; it.base_blk = 0; it.duo = 0;
        xq = RFL(xq); ticket = RFL(ticket);              // read from LDS: uniform, but only the hardware knows — keep the item in SGPRs
        if (ticket < 0) return it;
        const uint4* rp = reinterpret_cast<const uint4*>(a.qitems + ((long)xq * a.qcap + ticket));
        const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2];      // {duo, start, seg_end, qA} {soA, qB, soB, base_lo} {base_hi, -, -, -}
        it.live = 1; it.duo = RFL(r0.x); it.start = RFL((int)r0.y); it.seg_end = RFL((int)r0.z);
        it.qA = RFL((int)r0.w); it.soA = RFL((int)r1.x); it.qB = RFL((int)r1.y); it.soB = RFL((int)r1.z);
        it.base_blk = ((long)RFL((int)r2.x) << 32) | (unsigned)RFL((int)r1.w);
        return it;
    };
    auto chains_of = [&](const AdcItem& it, int ps) -> int {        // blocks wid, wid+16, ... of pass ps that exist
        const int pstart = it.start + ps * ADC_PASS_CODES;
        if (!it.live || pstart >= it.seg_end) return 0;
        const int nblk = (min(it.seg_end, pstart + ADC_PASS_CODES) - pstart + 63) >> 6;
        return (wid < ADC_WAVES && wid < nblk) ? (nblk - wid + ADC_WAVES - 1) / ADC_WAVES : 0;     // loader waves hold no chains
    };
    auto rsrc_of = [&](const AdcItem& it) -> adc_rsrc_t {           // the segment's first block
        const unsigned* b = a.codes + (it.base_blk + (it.start >> 6)) * (long)M4 * 64;
        return __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, 0x7FFFFFFF, 0x00020000);
    };
    auto off_of = [&](int ps) -> int { return (ps * (ADC_PASS_CODES >> 6) + wid) * M4 * 256; };    // the wave's first block of pass ps, bytes
    auto issue_table = [&](const AdcItem& it, int ph, int buf) {
        if constexpr (BUILD > 0) { adc_build_slab<BUILD>(a, it.live, it.duo, it.qA, it.qB, ph, reinterpret_cast<f32x2q*>(lds + (long)buf * (ADC_BUF_BYTES / 4)), wid, lane); return; }
        const float* __restrict__ src = a.lutg + (it.duo * n_ent + (long)ph * mp * KL) * 2;
        const int cnt = (min(M, (ph + 1) * mp) - ph * mp) * KL * 2;  // floats; a multiple of 4
        const unsigned dst = lds0 + (unsigned)buf * ADC_BUF_BYTES;
        constexpr int NL = ADC_LOADERS > 0 ? ADC_LOADERS : ADC_WAVES;              // the waves that move slabs: the loaders, or (ADC_LOADERS_N = 0) every wave
        const int lw = ADC_LOADERS > 0 ? wid - ADC_WAVES : wid;
        if (lw < 0) return;
        for (int e = lw * 256; e < cnt; e += NL * 256) {             // LDS-DMA: a wave moves 1 KiB per instruction, lane-linear destination
            if (e + (int)lane * 4 < cnt) adc_dma16(src + e, lane * 16u, dst + (unsigned)e * 4u);   // nt: a table is read by exactly one workgroup, once
        }
    };
    int my_q = blockIdx.x & 7, parity = 0;
    if (wid == 0) { const int t = take(my_q); if (lane == 0u) { s_ticket[0][0] = my_q; s_ticket[0][1] = t; } }
    __syncthreads();
    AdcItem cur = decode(s_ticket[0][0], s_ticket[0][1]);
    if (!cur.live) return;
    issue_table(cur, 0, 0);
    unsigned cw_cur[ADC_CHAINS][ADC_G];
    adc_rsrc_t rs_cur = rsrc_of(cur);
#pragma unroll
    for (int c = 0; c < ADC_CHAINS; c++)
#pragma unroll
        for (int i = 0; i < ADC_G; i++) cw_cur[c][i] = adc_ldw(rs_cur, voff, off_of(0) + c * stride16b + min(i, M4 - 1) * 256);
    int stage = 0;
#ifdef ADC_TRACE
    int trace_item = 0;
#endif
    while (true) {
        if (wid == 0) { const int t = take(my_q); if (lane == 0u) { s_ticket[parity ^ 1][0] = my_q; s_ticket[parity ^ 1][1] = t; } }   // next item's ticket
        AdcItem nxt = decode(0, -1);
        adc_rsrc_t rs_nxt = rs_cur;
        f32x2q acc[ADC_SEG_PASSES][ADC_CHAINS];
#pragma unroll
        for (int ps = 0; ps < ADC_SEG_PASSES; ps++)
#pragma unroll
            for (int c = 0; c < ADC_CHAINS; c++) { acc[ps][c][0] = 0.0f; acc[ps][c][1] = 0.0f; }
        const int na0 = chains_of(cur, 0), na1 = chains_of(cur, 1);
        ADC_STAMP(trace_item, 0);
        // Fused filter: the terms of a sum are squares, so a candidate's partial sum over the first phases never exceeds its sum
        // (float32 addition of non-negative terms is monotone). A wave whose candidates are ALL above their query's bound after a
        // phase cannot deliver a survivor: it skips the gathers of the item's remaining phases (the far lists of a query, mostly).
        // The bounds are read once per item, before the first phase (a stale bound is only looser).
        unsigned pTa = 0xFFFFFFFFu, pTb = 0xFFFFFFFFu;
        unsigned Tpre[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        bool dead = false;                                          // wave-uniform
        if (a.prune && na0 > 0) {                                   // (with the 4 ulp of slack of the epilogue's test; kept in SGPRs)
            auto slack = [](unsigned t) { return t >= 0x7F800000u ? 0xFFFFFFFFu : __float_as_uint(__uint_as_float(t) * 1.0000005f); };
            pTa = (unsigned)RFL((int)slack(__builtin_nontemporal_load(&a.tq[cur.qA])));
            pTb = cur.qB >= 0 ? (unsigned)RFL((int)slack(__builtin_nontemporal_load(&a.tq[cur.qB]))) : 0u;
        }
        for (int ph = 0; ph < P; ph++, stage++) {
            if constexpr (BUILD == 0) ADC_TABLE_WAIT();       // this wave's pieces of table (cur, ph) have landed ... (BUILD: its entries are written — the barrier waits for the LDS stores)
            __syncthreads();        // ... everybody's have; nobody still reads the other buffer; ticket visible
            ADC_STAMP(trace_item, 1 + 2 * ph);
            if (a.prune && ph > 0 && !dead && na0 > 0) {
                const unsigned sa = pTa, sb = pTb;
                bool alive = false;
#pragma unroll
                for (int ps = 0; ps < ADC_SEG_PASSES; ps++) {
                    const int na = ps == 0 ? na0 : na1;
#pragma unroll
                    for (int c = 0; c < ADC_CHAINS; c++) {
                        if (c >= na) continue;           // (lanes past the segment's end hold sums of foreign codes: they can only keep the wave alive)
                        alive = alive || __float_as_uint(acc[ps][c][0]) <= sa || __float_as_uint(acc[ps][c][1]) <= sb;
                    }
                }
                dead = __ballot(alive) == 0ull;
            }
            if (ph == 0) { nxt = decode(s_ticket[parity ^ 1][0], s_ticket[parity ^ 1][1]); if (nxt.live) rs_nxt = rsrc_of(nxt); }
            const bool last_ph = ph + 1 == P;
            if (last_ph && a.cand != nullptr && na0 > 0) {           // the filter's bounds, requested a phase ahead of their use (a round trip of ~2 k clocks per query otherwise)
                Tpre[0] = __builtin_nontemporal_load(&a.tq[cur.qA]);
                Tpre[1] = cur.qB >= 0 ? __builtin_nontemporal_load(&a.tq[cur.qB]) : 0u;
            }
            const bool build_late = BUILD > 0 && (wid & 1) == 0;            // BUILD: half of the waves form their share of the next slab AFTER their gathers — a build is a
                                                                            // chain of load round trips (residual slices, codewords), which then runs under the other half's gathers
            if (!build_late) {
                if (!last_ph) issue_table(cur, ph + 1, (stage + 1) & 1);
                else if (nxt.live) issue_table(nxt, 0, (stage + 1) & 1);
                ADC_STAMP(trace_item, 7 + ph);
            }
            const bool to_next = last_ph && nxt.live && chains_of(nxt, 0) > 0;   // the wave's next call belongs to the next item
            if (to_next && na0 == 0) {                               // idle in this item: fetch the next item's first code words now
#pragma unroll
                for (int c = 0; c < ADC_CHAINS; c++)
#pragma unroll
                    for (int i = 0; i < ADC_G; i++) cw_cur[c][i] = adc_ldw(rs_nxt, voff, off_of(0) + c * stride16b + min(i, M4 - 1) * 256);
            }
            const f32x2q* lut = reinterpret_cast<const f32x2q*>(lds + (long)(stage & 1) * (ADC_BUF_BYTES / 4));
            const int w_lo = (ph * mp) >> 2, m_hi = dead ? (w_lo << 2) : min(M, (ph + 1) * mp);   // a dead wave runs zero-width phases: no gathers, only the code words of its next call
            const int w_next = last_ph ? 0 : ((ph + 1) * mp) >> 2;   // first word of the wave's next call after this phase's last pass
#pragma unroll
            for (int ps = 0; ps < ADC_SEG_PASSES; ps++) {
                const int na = ps == 0 ? na0 : na1;
                if (na == 0) continue;
                const bool more = ps == 0 && na1 > 0;
                const adc_rsrc_t nrs = (!more && to_next) ? rs_nxt : rs_cur;
                const int nx_off = more ? off_of(1) : off_of(0);
                const int nxw = more ? w_lo : w_next;
                switch (na) {
                    case 1: adc_chains2<1>(lut, w_lo, m_hi, M4, KL, rs_cur, off_of(ps), stride16b, nrs, nx_off, nxw, voff, cw_cur, acc[ps]); break;
                    case 2: adc_chains2<2>(lut, w_lo, m_hi, M4, KL, rs_cur, off_of(ps), stride16b, nrs, nx_off, nxw, voff, cw_cur, acc[ps]); break;
                    case 3: adc_chains2<3>(lut, w_lo, m_hi, M4, KL, rs_cur, off_of(ps), stride16b, nrs, nx_off, nxw, voff, cw_cur, acc[ps]); break;
                    default: adc_chains2<4>(lut, w_lo, m_hi, M4, KL, rs_cur, off_of(ps), stride16b, nrs, nx_off, nxw, voff, cw_cur, acc[ps]); break;
                }
            }
            ADC_STAMP(trace_item, 2 + 2 * ph);
            if (build_late) {
                if (!last_ph) issue_table(cur, ph + 1, (stage + 1) & 1);
                else if (nxt.live) issue_table(nxt, 0, (stage + 1) & 1);
                ADC_STAMP(trace_item, 7 + ph);
            }
        }
        if (a.cand == nullptr) {
#pragma unroll
            for (int ps = 0; ps < ADC_SEG_PASSES; ps++) {
                const int na = ps == 0 ? na0 : na1;
#pragma unroll
                for (int c = 0; c < ADC_CHAINS; c++) {
                    const int j = ((cur.start >> 6) + ps * (ADC_PASS_CODES >> 6) + c * ADC_WAVES + wid) * 64 + (int)lane;
                    if (c < na && j < cur.seg_end) {
                        const bool ok = a.elig ? (a.elig[(cur.base_blk << 6) + j] != 0) : true;
                        a.D[(long)cur.qA * a.ldD + cur.soA + j] = ok ? go_sqrt32q(acc[ps][c][0]) : __uint_as_float(EXCLUDED_BITS);
                        if (cur.qB >= 0) a.D[(long)cur.qB * a.ldD + cur.soB + j] = ok ? go_sqrt32q(acc[ps][c][1]) : __uint_as_float(EXCLUDED_BITS);
                    }
                }
            }
        } else if (na0 > 0) {
            // Fused top-K filter (ivfpq_index_search.go:310-321 keeps every candidate and sorts; only the K best matter). tq[q] is an
            // upper bound on the K-th smallest SUM of the query (the square root is monotone, so its root bounds the K-th smallest
            // distance): the K-th smallest of ANY K+ of a query's candidates is one, so every wave offers the K-th smallest of its
            // lanes' minima and applies it to its own candidates at once; the shared bound only ever tightens (atomicMin), and a
            // candidate above the bound it was tested against can never be among the K best, whichever bound that was. Most items
            // (the far lists) have nothing under the bound and leave after eight compares — without taking a single square root.
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int q = h ? cur.qB : cur.qA, so = h ? cur.soB : cur.soA;
                if (q < 0) continue;
                const unsigned T = Tpre[h];                                                // float bits of a sum >= 0: unsigned order = value order (read during the last phase: a stale bound is only looser)
                const unsigned Ts = __float_as_uint(__uint_as_float(T) * 1.0000005f);     // sums within 4 ulp above the bound may round to the same distance
#ifdef ADC_TRACE
                if (Ts == 12345u) asm volatile("s_nop 0");       // force the wait for the bound before the stamp
                ADC_STAMP(trace_item, 10 + 2 * h);
#endif
                unsigned keys[ADC_SEG_PASSES][ADC_CHAINS];
                unsigned lmin = 0xFFFFFFFFu;
#pragma unroll
                for (int ps = 0; ps < ADC_SEG_PASSES; ps++) {
                    const int na = ps == 0 ? na0 : na1;
#pragma unroll
                    for (int c = 0; c < ADC_CHAINS; c++) {
                        keys[ps][c] = 0xFFFFFFFFu;
                        if (c >= na) continue;                               // wave-uniform
                        const int j = ((cur.start >> 6) + ps * (ADC_PASS_CODES >> 6) + c * ADC_WAVES + wid) * 64 + (int)lane;
                        bool ok = j < cur.seg_end;
                        if (ok && a.elig) ok = a.elig[(cur.base_blk << 6) + j] != 0;     // soft-deleted / filtered candidates never count
                        if (ok) { keys[ps][c] = __float_as_uint(acc[ps][c][h]); lmin = min(lmin, keys[ps][c]); }
                    }
                }
                ADC_STAMP(trace_item, 11 + 2 * h);
                // waves of the item that hold chains: every one of them "arrives" once per half, the last one flushes the staged survivors
                const int nblk0 = (min(cur.seg_end, cur.start + ADC_PASS_CODES) - cur.start + 63) >> 6;
                const int nw = min(ADC_WAVES, nblk0);
                if (__ballot(lmin <= Ts) != 0ull) {                          // else: nothing of this wave can matter (and its K-th minimum is above the bound)
                // K-th smallest lane minimum, bit by bit from the top (ballots only: the LDS pipe is the kernel's bottleneck)
                unsigned kth = 0xFFFFFFFFu;
                if ((int)__builtin_popcountll(__ballot(lmin != 0xFFFFFFFFu)) >= a.K) {
                    kth = 0u;
#pragma unroll
                    for (int bit = 31; bit >= 0; bit--) {
                        const unsigned tv = kth | ((1u << bit) - 1u);
                        if ((int)__builtin_popcountll(__ballot(lmin <= tv)) < a.K) kth |= 1u << bit;
                    }
                }
                // (the bound goes to the query's word through the item's flushing wave: tq[] is a dense array — 32 queries share a 128-byte line — and a global
                // atomic per wave from every workgroup of a launch serialised on a handful of lines: the first item of a B = 32 launch took 0.4 of its 0.54 ms)
                if (lane == 0 && kth < T) atomicMin(&s_kmin[h], kth);
                const unsigned bnd = min(T, kth);
                const unsigned bs = bnd >= 0x7F800000u ? 0x7F800000u : __float_as_uint(__uint_as_float(bnd) * 1.0000005f);
                const float Td = go_sqrt32q(__uint_as_float(bnd));           // the bound as a distance
#pragma unroll
                for (int ps = 0; ps < ADC_SEG_PASSES; ps++) {
                    const int na = ps == 0 ? na0 : na1;
#pragma unroll
                    for (int c = 0; c < ADC_CHAINS; c++) {
                        if (c >= na) continue;
                        bool keep = keys[ps][c] <= bs;                       // (excluded candidates are 0xFFFFFFFF)
                        if (__ballot(keep) == 0ull) continue;
                        const float d = go_sqrt32q(acc[ps][c][h]);
                        keep = keep && d <= Td && !(a.thr > 0.0f && d > a.thr);          // `s.threshold > 0 && dist > s.threshold`
                        const unsigned long long m = __ballot(keep);
                        if (m) {
                            const int j = ((cur.start >> 6) + ps * (ADC_PASS_CODES >> 6) + c * ADC_WAVES + wid) * 64 + (int)lane;
                            const int leader = __builtin_ctzll(m);
                            const int cnt = (int)__builtin_popcountll(m);
                            const unsigned long long comp = ((unsigned long long)adc_f2key(__float_as_uint(d)) << 32) | (unsigned)(so + j);
                            const int rank = (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                            int slot = 0;
                            if ((int)lane == leader) slot = atomicAdd(&s_scnt[h], cnt);                    // LDS: a slot range of the item's staging area
                            slot = RFL(__shfl(slot, leader, 64));
                            if (slot + cnt <= ADC_STAGE_CAP) { if (keep) s_stage[h][slot + rank] = comp; }
                            else {                                                                        // the staging area is full: append directly (rare)
                                if ((int)lane == leader) atomicMin(&s_svalid[h], slot);
                                int base = 0;
                                if ((int)lane == leader) base = atomicAdd(&a.cursor[q], cnt);
                                base = RFL(__shfl(base, leader, 64));
                                if (keep) __hip_atomic_store(&a.cand[(long)q * a.ldD + base + rank], comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
                }
                // arrival: the last wave of the item to get here appends what was staged for this half — ONE returning atomic on the query's cursor
                __threadfence_block();
                int arrived = 0;
                if (lane == 0) arrived = atomicAdd(&s_sdone[h], 1);
                arrived = RFL(__shfl(arrived, 0, 64));
                if (arrived == nw - 1) {
                    __threadfence_block();
                    const int n = min(min(s_scnt[h], s_svalid[h]), ADC_STAGE_CAP);
                    int app_lo = 0, app_hi = 0;
                    if (n > 0) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&a.cursor[q], n);
                        base = RFL(__shfl(base, 0, 64));
                        // (agent-scope stores: written through the XCD's L2, so that a refining wave on another XCD can see them)
                        for (int i = (int)lane; i < n; i += 64) __hip_atomic_store(&a.cand[(long)q * a.ldD + base + i], s_stage[h][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        app_lo = base; app_hi = base + n;
                    }
                    const unsigned kmin = s_kmin[h];
                    if (lane == 0 && kmin != 0xFFFFFFFFu) atomicMin(&a.tq[q], kmin);
                    if (lane == 0) { s_scnt[h] = 0; s_sdone[h] = 0; s_svalid[h] = 0x7FFFFFFF; s_kmin[h] = 0xFFFFFFFFu; }           // for the next item (three phase barriers away)
                    // Bound refinement from the survivors. A wave can only offer the K-th smallest of ITS lanes' minima — of 64 .. 512 candidates; on data without cluster
                    // structure (SURVEY 8d's uniform rows) that is the ~K / 100-quantile of a query's distances, every wave of every later item still has candidates under
                    // it, and every item runs the slow path above. The query's survivor row, though, holds every candidate that passed so far: the K-th smallest DISTANCE
                    // among any of them is the K-th best of a subset of the query's candidates, hence an upper bound on the K-th best of all. The flushing wave whose append
                    // crosses a power of two (16, 32, ... ADC_REFINE_MAX) reads the row's first entries (unwritten slots read as the row's initial all-ones = +inf: looser,
                    // never wrong), finds their K-th smallest key bit by bit and lowers tq to the smallest sum bound that cannot cut a candidate at that distance: the
                    // correctly rounded root of S is D  =>  S <= D^2 (1 + 2^-23), and the test against tq carries its own 4 ulp.
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
                            const float Dk = __uint_as_float((kk & 0x80000000u) ? (kk & 0x7FFFFFFFu) : ~kk);      // the key back to the distance (adc_f2key's inverse)
                            const float Sb = (Dk * Dk) * 1.000001f;
                            if (lane == 0 && Sb == Sb && __float_as_uint(Sb) < 0x7F800000u) atomicMin(&a.tq[q], __float_as_uint(Sb));
                        }
                    }
                }
            }
        }
        ADC_STAMP(trace_item, 15);
        if (!nxt.live) break;
        cur = nxt; rs_cur = rs_nxt; parity ^= 1;
#ifdef ADC_TRACE
        trace_item++;
#endif
    }
}
#include "kernels_adc2.inc.hpp"
// pairs of a (sub-)batch grouped by probed list; falls back to the identity order when the list count does not fit in LDS
bool launch_order_pairs(Ctx* c, const uint32_t* probe_list, int ldp, int np, const int32_t* seg_off, int n_pairs, int nlist, uint32_t* order, uint32_t* olist) {
    if (n_pairs <= 0) return false;
    static bool attr_done = false;
    if (!attr_done) { HIP_CHECK(hipFuncSetAttribute((const void*)order_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (ORDER_MAX_LISTS + 1) * 4)); attr_done = true; }
    ProfScope ps(c, "order_pairs");
    const bool sorted = nlist <= ORDER_MAX_LISTS;
    if (sorted)
        order_pairs_kernel<<<dim3(1), dim3(1024), (size_t)(nlist + 1) * 4, c->stream>>>(probe_list, ldp, np, seg_off, n_pairs, nlist, order, olist);
    else
        iota_kernel<<<dim3((unsigned)ceil_div(n_pairs, 256)), dim3(256), 0, c->stream>>>(order, n_pairs);
    LAUNCH_CHECK();
    return sorted;     // false: identity order, olist not written (the caller must not group by list)
}
int64_t adc_codes_pad() { return (int64_t)ADC_PASS_CODES; }
size_t adc_lds_bytes(int M, int Ksub, int dim) { (void)M; (void)Ksub; (void)dim; return 2 * (size_t)ADC_BUF_BYTES; }
static size_t adc_lut_budget() {
    static size_t b = [] { const char* e = getenv("COMET_ADC_LUT_MB"); long mb = e ? atol(e) : 1024; if (mb < 1) mb = 1; return (size_t)mb << 20; }();
    return b;
}
// Sub-batching of launch_adc_scan and whether it runs the two-stage search: functions of (M, Ksub, np, B, nlist, the filter's mode) only — the same on
// every rank of a sharded search, whatever the rank's own lists hold. The stage-1 bound exchange (one collective per sub-batch) relies on that.
static int64_t adc_sub_batch(int M, int Ksub, int np, int B) {
    const int KL = Ksub < 256 ? Ksub : 256;
    const size_t lut_pair = (size_t)M * KL * sizeof(float);
    const int64_t qc = std::max<int64_t>(1, (int64_t)(adc_lut_budget() / (lut_pair * (size_t)np)));
    return std::min<int64_t>(qc, B);
}
static bool adc_two_stage(const AdcFilter* flt, int np, int nlist) {
    static const bool two_stage_on = getenv("COMET_ADC_ONE_STAGE") == nullptr;
    return flt != nullptr && np >= 2 && nlist <= ORDER_MAX_LISTS && two_stage_on && !flt->one_stage;
}
// A rank of a sharded search that has nothing to scan (an empty shard) still takes part in every exchange its peers issue: +inf bounds, the same
// sub-batches. `tq` must hold B words.
void adc_exchange_idle(Ctx* c, const AdcFilter* flt, int M, int Ksub, int np, int B, int nlist) {
    if (!flt || !flt->exchange || B <= 0 || np <= 0 || !adc_two_stage(flt, np, nlist)) return;
    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)flt->tq, 0x7F800000, (size_t)B, c->stream));
    const int qc = (int)adc_sub_batch(M, Ksub, np, B);
    for (int b0 = 0; b0 < B; b0 += qc) flt->exchange(flt->exchange_user, flt->tq + b0, std::min(qc, B - b0));
}
int adc_exchange_plan(int M, int Ksub, int np, int B, int nlist, int* per) {
    AdcFilter two{}; two.one_stage = 0;
    if (B <= 0 || np <= 0 || !adc_two_stage(&two, np, nlist)) { if (per) *per = 0; return 0; }
    const int qc = (int)adc_sub_batch(M, Ksub, np, B);
    if (per) *per = qc;
    return (B + qc - 1) / qc;
}
void launch_adc_scan(Ctx* c, const float* Qp, int ld, int dim, const float* centroids, const float* codebooks, int M, int Ksub, int dsub,
                     const uint32_t* codes, int M4, const int64_t* list_base, const int32_t* list_len, const uint32_t* probe_list, int ldp,
                     int np, const int32_t* seg_off, const uint8_t* elig, int B, int nlist, int max_list_len, float* D, int64_t ldD, const AdcFilter* flt, int64_t n_codes) {
    if (B <= 0 || np <= 0) return;
    if (max_list_len <= 0) { adc_exchange_idle(c, flt, M, Ksub, np, B, nlist); return; }
    const int KL = Ksub < 256 ? Ksub : 256;
    const int kl_shift = 31 - __builtin_clz((unsigned)KL);
    const size_t lds = adc_lds_bytes(M, Ksub, dim);
    const int mp = std::max(8, (int)(ADC_BUF_BYTES / ((size_t)KL * 8)) / 8 * 8);           // subspaces per phase buffer: whole groups of ADC_G = 2 code words (KL <= 256 and 64 KiB buffers: >= 32)
    const bool identity = nlist > ORDER_MAX_LISTS;
    // queries per sub-batch: tables of a sub-batch live in HBM between the two kernels
    const int64_t qc = adc_sub_batch(M, Ksub, np, B);
    const bool lead = flt != nullptr && np >= 2 && !identity;
    auto slots_for = [&](int64_t n_pairs) { return identity ? 2 * n_pairs : (lead ? 2 * (n_pairs / np) : 0) + round_up(n_pairs + std::min<int64_t>(nlist, n_pairs), 2); };
    const int64_t max_slots = slots_for(qc * np);
    // 8-bit codebooks with 4 / 8 / 16 dimensions per subspace: the scanning workgroups build the tables themselves, in LDS (adc_scan_kernel<DSUB>): no table kernel,
    // no table bytes in HBM. COMET_ADC_STREAM_TABLES=1 keeps the streamed form (pq_lut_kernel -> HBM -> LDS-DMA), which every other shape uses.
    static const bool stream_tables = getenv("COMET_ADC_STREAM_TABLES") != nullptr;
    const int build = (!stream_tables && KL == 256 && (dsub == 4 || dsub == 8) && (size_t)M * dsub <= (size_t)ld && mp * KL * 8 <= ADC_BUF_BYTES) ? dsub : 0;
    // round 6: the batched, phase-major form (adc_scan2_kernel: codeword slices in registers, A2_G items per slice, batches software-pipelined) for the same
    // shapes. It wins where a launch is many items of a few thousand codes (configs[3]'s shape: 0.506 -> 0.460 ms for the every-candidate scan) and loses on
    // short or very uneven lists and on the small launches of the two-stage search (a batch of four items per workgroup is a longer critical path than four
    // workgroups with one item each) — so: single-stage launches of indexes whose average list holds >= 1536 codes. COMET_ADC_KERNEL=1 / 2 forces a kernel
    // (read per search: the tests run both forms in one process).
    const char* kenv = getenv("COMET_ADC_KERNEL");
    const int kforce = kenv ? atoi(kenv) : 0;
    const bool scan2_ok = build != 0 && Ksub == 256 && M * dsub <= A2_RES_DIMS;
    const bool scan2_auto = !adc_two_stage(flt, np, nlist) && n_codes / std::max(1, nlist) >= 1536;
    const bool scan2 = scan2_ok && (kforce == 2 || (kforce != 1 && scan2_auto));
    const int seg_codes = scan2 ? A2_SEG_CODES : ADC_SEG_CODES;
    const int segs = (int)ceil_div(max_list_len, seg_codes);
    const int64_t max_chunks = ceil_div(max_slots / 2, ADC_XCD_CHUNK);
    const int64_t qcap = ceil_div(max_chunks, 8) * ADC_XCD_CHUNK * segs;
    if (qcap > (int64_t)1 << 28) COMET_FAIL(COMET_ERR_UNSUPPORTED, "ADC work queue too large (%lld items)", (long long)qcap);
    ScratchMark mark(c);
    float* lut = build ? nullptr : c->salloc<float>((size_t)max_slots * M * KL);
    float* zero_row = nullptr;                                       // PQ has no coarse centroid: its "residual" is the query minus a row of zeros
    if (build && !centroids) { zero_row = c->salloc<float>((size_t)ld); c->zero(zero_row, (size_t)ld * 4); }
    uint32_t* order = c->salloc<uint32_t>((size_t)max_slots);
    uint32_t* slist = c->salloc<uint32_t>((size_t)max_slots);
    AdcRec* qitems = c->salloc<AdcRec>((size_t)8 * qcap);
    int32_t* qcount = c->salloc<int32_t>(8);
    int32_t* queues = c->salloc<int32_t>(8);
    const int mw = 256 >> kl_shift;
    int ppw = LUT_PAIRS_PER_WG;
    while (ppw > 2 && (size_t)ppw * mw * dsub * 4 > 48 * 1024) ppw >>= 1;
    const size_t lut_lds = (size_t)ppw * mw * dsub * 4 + (size_t)ppw * 4;
    if (lut_lds > 150 * 1024) COMET_FAIL(COMET_ERR_UNSUPPORTED, "PQ subspace slice too wide for the table-build kernel (%zu bytes of LDS)", lut_lds);
    static bool attr_done = false;
    if (!attr_done) {
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (ORDER_MAX_LISTS + 1) * 4));
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A2_LDS_BYTES));
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan2_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A2_LDS_BYTES));
        attr_done = true;
    }
    // Two-stage fused search (K <= 64, np >= 2): stage 1 scans every query's NEAREST list only, which seeds the per-query bounds; a
    // lower bound per remaining (query, list) pair — the serial sum of its table's row minima, pq_rowmin_kernel / pq_lb_kernel — then
    // removes every pair none of whose candidates can pass the bound (no table, no slot, no item), and stage 2 scans what is left.
    // Exact: a removed candidate's sum is above a bound that only ever tightens. On clustered data almost everything behind the
    // nearest lists goes (bench corpus: 98.6 % of the candidates); on unclustered data the cost is the row-minima kernel.
    const bool two_stage = adc_two_stage(flt, np, nlist);
    const int strict = (two_stage && flt->exchange) ? 1 : 0;     // sharded search with a bound exchange: stage 1 = probe 0 on its owner only (adc_is_first)
    int32_t* used = c->salloc<int32_t>(4);
    float* rowmin = two_stage ? c->salloc<float>((size_t)qc * np * M) : nullptr;
    uint8_t* dead = two_stage ? c->salloc<uint8_t>((size_t)qc * np) : nullptr;
    const size_t rm_lds = lut_lds + (size_t)ppw * mw * (KL > 64 ? KL / 64 : 1) * 4;
    for (int b0 = 0; b0 < B; b0 += (int)qc) {
        const int bn = std::min<int>((int)qc, B - b0);
        const int n_pairs = bn * np;
        const float* Qb = Qp + (size_t)b0 * ld;
        const uint32_t* pl = probe_list + (size_t)b0 * ldp;
        const int32_t* so = seg_off + (size_t)b0 * (np + 1);
        auto run_stage = [&](int stage) {
            const int n_slots = (int)(stage == 0 ? slots_for(n_pairs) : round_up((stage == 1 ? bn : (strict ? n_pairs : n_pairs - bn)) + std::min<int64_t>(nlist, n_pairs), 2));
            {
                ProfScope ps(c, "adc_order");
                const bool init_rows = flt && flt->cand && stage <= 1;
                adc_order_kernel<<<dim3(init_rows ? 1 + (unsigned)std::min(bn, 64) : 1), dim3(1024), identity ? 0 : (size_t)(nlist + 1) * 4, c->stream>>>(pl, ldp, np, so, n_pairs, nlist, identity ? 1 : 0, list_len,
                                                                                                           n_slots, order, slist, qitems, (int)qcap, qcount, queues,
                                                                                                           flt ? flt->tq + b0 : nullptr, flt ? flt->cursor + b0 : nullptr, bn, (lead && stage == 0) ? 1 : 0, (const long*)list_base,
                                                                                                           stage, dead, used, flt ? flt->stats : nullptr, strict,
                                                                                                           init_rows ? flt->cand + (size_t)b0 * ldD : nullptr, (long)ldD, seg_codes, scan2 ? 1 : 0);
                LAUNCH_CHECK();
            }
            if (!build) {
                ProfScope ps(c, "pq_lut");
                dim3 grid((unsigned)ceil_div(M, mw), (unsigned)ceil_div(n_slots, ppw)), blk(256);
#define LUT_LAUNCH(HC, DS) do { if (lut_lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)pq_lut_kernel<HC, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lut_lds)); \
        pq_lut_kernel<HC, DS><<<grid, blk, lut_lds, c->stream>>>(Qb, ld, centroids, codebooks, M, Ksub, KL, kl_shift, dsub, pl, ldp, np, so, order, n_slots, ppw, lut, used, (stage == 0 && (size_t)n_slots * M * KL * 4 > ((size_t)128 << 20)) ? 1 : 0); } while (0)
#define LUT_DS(HC) do { switch (dsub) { case 2: LUT_LAUNCH(HC, 2); break; case 4: LUT_LAUNCH(HC, 4); break; case 8: LUT_LAUNCH(HC, 8); break; \
                                       case 16: LUT_LAUNCH(HC, 16); break; default: LUT_LAUNCH(HC, 0); break; } } while (0)
                if (centroids) LUT_DS(true); else LUT_DS(false);
#undef LUT_DS
#undef LUT_LAUNCH
                LAUNCH_CHECK();
            }
            {   // (timed through Ctx::launch_timed: no event-record packets around the kernel when a bench times this scope)
                const long n_items = (long)(n_slots / 2) * segs;
                long g = std::min<long>(n_items, (long)c->prop.multiProcessorCount * std::min(160 / (2 * ADC_BUF_KB + 2), ADC_WPE * 256 / ADC_THREADS));    // as many workgroups per CU as their LDS and registers allow
                g = std::max<long>(8, (g + 7) / 8 * 8);          // a multiple of the XCD count so that blockIdx % 8 is the XCD of every slot
                static const bool prune_on = getenv("COMET_ADC_NO_PRUNE") == nullptr;
                static const bool refine_on = getenv("COMET_ADC_NO_REFINE") == nullptr;
                AdcArgs a{lut, codes, (const long*)list_base, list_len, so, elig, order, slist, qitems, qcount, queues, D ? D + (size_t)b0 * ldD : nullptr, (long)ldD, M, KL, mp, M4, np, (int)qcap,
                          flt ? flt->cand + (size_t)b0 * ldD : nullptr, flt ? flt->cursor + b0 : nullptr, flt ? flt->tq + b0 : nullptr, flt ? flt->K : 0, flt ? flt->thr : 0.0f, (flt && prune_on) ? 1 : 0, (flt && refine_on) ? 1 : 0,
                          Qb, centroids ? centroids : zero_row, codebooks, ld, Ksub, centroids ? 1 : 0};
                if (scan2) {
                    long g2 = std::min<long>(n_items, (long)c->prop.multiProcessorCount);     // one 8-wave workgroup of 256 registers per CU
                    g2 = std::max<long>(8, (g2 + 7) / 8 * 8);
                    if (build == 4) c->launch_timed("adc_scan", adc_scan2_kernel<4>, dim3((unsigned)g2), dim3(A2_THREADS), (size_t)A2_LDS_BYTES, a);
                    else c->launch_timed("adc_scan", adc_scan2_kernel<8>, dim3((unsigned)g2), dim3(A2_THREADS), (size_t)A2_LDS_BYTES, a);
                } else
                switch (build) {
                    case 4: c->launch_timed("adc_scan", adc_scan_kernel<4>, dim3((unsigned)g), dim3(ADC_THREADS), lds, a); break;
                    case 8: c->launch_timed("adc_scan", adc_scan_kernel<8>, dim3((unsigned)g), dim3(ADC_THREADS), lds, a); break;
                    case 16: c->launch_timed("adc_scan", adc_scan_kernel<16>, dim3((unsigned)g), dim3(ADC_THREADS), lds, a); break;
                    default: c->launch_timed("adc_scan", adc_scan_kernel<0>, dim3((unsigned)g), dim3(ADC_THREADS), lds, a); break;
                }
                LAUNCH_CHECK();
            }
        };
        if (!two_stage) { run_stage(0); continue; }
        run_stage(1);
        if (flt->exchange) flt->exchange(flt->exchange_user, flt->tq + b0, bn);     // list shards: the global bound (one small all-reduce per sub-batch)
        static const bool bound_off = getenv("COMET_ADC_ROWMIN") != nullptr;      // the two-kernel form (row minima of every subspace, then the sums)
        // 8-bit codebooks, DSUB 4 / 8 / 16: the survivors of the table-free tests are compacted (pq_bound_prep_kernel) and bounded without a walk (pq_bound3_kernel)
        if (!bound_off && flt->bound_tab3 && KL == 256 && Ksub == 256 && (dsub == 4 || dsub == 8 || dsub == 16) && (size_t)M * dsub <= (size_t)ld) {
            {
            ProfScope ps(c, "pq_bound");
            unsigned* plist = c->salloc<unsigned>((size_t)n_pairs + 8);
            int* pcount = c->salloc<int>(4);
            HIP_CHECK(hipMemsetAsync(pcount, 0, 4, c->stream));
            const dim3 pg((unsigned)ceil_div(n_pairs, PREP_WAVES));
            if (centroids) pq_bound_prep_kernel<true><<<pg, dim3(PREP_WAVES * 64), 0, c->stream>>>(Qb, ld, centroids, M * dsub, pl, ldp, np, so, n_pairs, flt->tq + b0, dead, flt->stats, flt->list_rmax, strict, plist, pcount);
            else pq_bound_prep_kernel<false><<<pg, dim3(PREP_WAVES * 64), 0, c->stream>>>(Qb, ld, centroids, M * dsub, pl, ldp, np, so, n_pairs, flt->tq + b0, dead, flt->stats, flt->list_rmax, strict, plist, pcount);
#define BND_LAUNCH(HC, DS) pq_bound3_kernel<HC, DS, BND3_P><<<dim3((unsigned)ceil_div(n_pairs, BND3_P)), dim3(256), 0, c->stream>>>(Qb, ld, centroids, flt->bound_tab3, flt->bound_cmax2, M, pl, ldp, np, flt->tq + b0, dead, flt->stats, plist, pcount)
#define BND_DS(HC) do { switch (dsub) { case 4: BND_LAUNCH(HC, 4); break; case 8: BND_LAUNCH(HC, 8); break; default: BND_LAUNCH(HC, 16); break; } } while (0)
            if (centroids) BND_DS(true); else BND_DS(false);
#undef BND_DS
#undef BND_LAUNCH
            LAUNCH_CHECK();
            }
            run_stage(2);
            continue;
        }
        {
            ProfScope ps(c, "pq_rowmin");
            dim3 grid((unsigned)ceil_div(M, mw), (unsigned)ceil_div(n_pairs, ppw)), blk(256);
#define RM_LAUNCH(HC, DS) do { if (rm_lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)pq_rowmin_kernel<HC, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rm_lds)); \
        pq_rowmin_kernel<HC, DS><<<grid, blk, rm_lds, c->stream>>>(Qb, ld, centroids, codebooks, M, Ksub, KL, kl_shift, dsub, pl, ldp, np, so, n_pairs, ppw, rowmin, strict); } while (0)
#define RM_DS(HC) do { switch (dsub) { case 2: RM_LAUNCH(HC, 2); break; case 4: RM_LAUNCH(HC, 4); break; case 8: RM_LAUNCH(HC, 8); break; \
                                      case 16: RM_LAUNCH(HC, 16); break; default: RM_LAUNCH(HC, 0); break; } } while (0)
            if (centroids) RM_DS(true); else RM_DS(false);
#undef RM_DS
#undef RM_LAUNCH
            LAUNCH_CHECK();
        }
        {
            ProfScope ps(c, "pq_lb");
            if (64 * (M + 1) * 4 > 64 * 1024) COMET_FAIL(COMET_ERR_UNSUPPORTED, "too many PQ subspaces for the lower-bound kernel (%d)", M);
            pq_lb_kernel<<<dim3((unsigned)ceil_div(n_pairs, 64)), dim3(256), (size_t)64 * (M + 1) * 4, c->stream>>>(rowmin, M, n_pairs, np, so, flt->tq + b0, dead, flt->stats, strict);
            LAUNCH_CHECK();
        }
        run_stage(2);
    }
}

}  // namespace comet

#ifdef A2_TRACE
extern "C" __attribute__((visibility("default"))) int comet_debug_a2_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(comet::a2_trace_buf), (size_t)n * 8);
}
#endif
#ifdef ADC_TRACE
extern "C" __attribute__((visibility("default"))) int comet_debug_adc_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(comet::adc_trace_buf), (size_t)n * 8);
}
#endif
