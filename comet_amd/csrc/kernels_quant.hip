// kernels_quant.hip — k-means, coarse assignment, PQ encode and the fused PQ lookup-table + ADC scan
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
    for (int i = 0; i < cnt; i++) sum = sum + V[(long)mem[i] * ld + col];       // clusterSums[c][dim] += v[dim]
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
    km_update_kernel<<<dim3((unsigned)ceil_div(ld, 256), k), dim3(256), 0, c->stream>>>(V, ld, members, offs, counts, k, centroids); LAUNCH_CHECK();
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
// probe bookkeeping shared by IVF and IVFPQ
// ------------------------------------------------------------------------------------------------
// seg_off[q][0..np] = prefix of list_len over the probed lists; cnts[q] = total candidates
__global__ __launch_bounds__(64) void probe_segments_kernel(const unsigned* __restrict__ probe_list, int ldp, const int* __restrict__ probe_cnt,
                                                            const int* __restrict__ list_len, int B, int np, int* __restrict__ seg_off,
                                                            int* __restrict__ cnts) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B) return;
    int run = 0;
    const int npq = probe_cnt ? probe_cnt[q] : np;
    for (int p = 0; p < np; p++) {
        seg_off[(long)q * (np + 1) + p] = run;
        if (p < npq) run += list_len[probe_list[(long)q * ldp + p]];
    }
    seg_off[(long)q * (np + 1) + np] = run;
    cnts[q] = run;
}
void launch_probe_segments(Ctx* c, const uint32_t* probe_list, int ldp, const int32_t* probe_cnt, const int32_t* list_len, int B, int np,
                           int32_t* seg_off, int32_t* cnts) {
    probe_segments_kernel<<<dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, c->stream>>>(probe_list, ldp, probe_cnt, list_len, B, np, seg_off, cnts);
    LAUNCH_CHECK();
}
__device__ __forceinline__ int find_probe(const int* __restrict__ so, int np, int pos) {
    // largest p with so[p] <= pos (so non-decreasing, so[np] = total); skips empty lists
    int lo = 0, hi = np;   // invariant: so[lo] <= pos < so[hi]
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (so[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}
// rowidx[q][pos] = row_of_slot[list_base[L] + j] (or EXCLUDED when the slot is ineligible / pos >= cnt)
__global__ __launch_bounds__(256) void cand_rows_kernel(const unsigned* __restrict__ probe_list, int ldp, const int* __restrict__ seg_off, int np,
                                                        const long* __restrict__ list_base, const unsigned* __restrict__ row_of_slot,
                                                        const unsigned char* __restrict__ elig, const int* __restrict__ cnts,
                                                        unsigned* __restrict__ rowidx, long ldR) {
    const int q = blockIdx.y;
    const long pos = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= ldR) return;
    unsigned out = 0xFFFFFFFFu;
    if (pos < cnts[q]) {
        const int* so = seg_off + (long)q * (np + 1);
        int p = find_probe(so, np, (int)pos);
        long slot = list_base[probe_list[(long)q * ldp + p]] + (pos - so[p]);
        if (!elig || elig[slot]) out = row_of_slot ? row_of_slot[slot] : (unsigned)slot;
    }
    rowidx[(long)q * ldR + pos] = out;
}
void launch_cand_rows(Ctx* c, const uint32_t* probe_list, int ldp, const int32_t* seg_off, int np, const int64_t* list_base,
                      const uint32_t* row_of_slot, const uint8_t* elig, const int32_t* cnts, int B, uint32_t* rowidx, int64_t ldR) {
    if (B <= 0 || ldR <= 0) return;
    ProfScope ps(c, "cand_rows");
    cand_rows_kernel<<<dim3((unsigned)ceil_div(ldR, 256), B), dim3(256), 0, c->stream>>>(probe_list, ldp, seg_off, np, (const long*)list_base, row_of_slot, elig, cnts, rowidx, ldR);
    LAUNCH_CHECK();
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
// fused PQ lookup-table build + asymmetric-distance scan (pq_index_search.go:243-306,
// ivfpq_index_search.go:285-321,350-390).
//
// One workgroup = one (query, probed list, segment of that list). It
//   1. forms the query residual r = q - centroid[list] in LDS (IVFPQ; r = q for plain PQ),
//   2. builds the M x KL table LUT[m][k] = sum_i (r[m*dsub+i] - cb[m][k][i])^2 in LDS (KL = min(Ksub,256):
//      codes are uint8, so entries >= 256 can never be addressed), exact float32, dimension order,
//   3. streams the list's codes — stored as 64-code blocks, word-interleaved so that a wave reads
//      256 contiguous bytes per load — and for each code sums LUT[m][code[m]] in m order (the
//      reference's serial float32 sum), takes the correctly-rounded sqrt and writes the distance.
// LDS holds the table (96 KiB at M=96, Ksub=256: fits because gfx950 has 160 KiB per CU).
// ------------------------------------------------------------------------------------------------
constexpr int ADC_THREADS = 1024;
typedef float f32x4q __attribute__((ext_vector_type(4)));

template <bool HAS_CENTROID, int DSUB>
__global__ __launch_bounds__(ADC_THREADS) void adc_scan_kernel(const float* __restrict__ Qp, int ld, int dim,
                                                               const float* __restrict__ centroids, const float* __restrict__ codebooks,
                                                               int M, int Ksub, int KL, int dsub, const unsigned* __restrict__ codes, int M4,
                                                               const long* __restrict__ list_base, const int* __restrict__ list_len,
                                                               const unsigned* __restrict__ probe_list, int ldp, int np,
                                                               const int* __restrict__ seg_off, const unsigned char* __restrict__ elig,
                                                               float* __restrict__ D, long ldD, int segs_per_list, int seg_codes, int n_queries) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lut = lds;               // M * KL
    float* res = lds + (long)M * KL;  // dim
    // persistent workgroups: the 96 KiB table allows one workgroup per CU, so re-launching one per work item
    // would expose the dispatch gap ~30 times per CU; instead each resident workgroup walks the item list
    // (query, probe, segment) with a grid stride.
    const int items_per_q = np * segs_per_list;
    const long n_items = (long)n_queries * items_per_q;
  for (long item = blockIdx.x; item < n_items; item += gridDim.x) {
    __syncthreads();                 // previous item's table no longer in use
    const int q = (int)(item / items_per_q);
    const int rem = (int)(item - (long)q * items_per_q);
    const int p = rem / segs_per_list;
    const int s = rem - p * segs_per_list;
    const int so = seg_off[(long)q * (np + 1) + p];
    if (seg_off[(long)q * (np + 1) + p + 1] == so) continue;   // empty list or probe slot not used by this query
    const unsigned L = probe_list[(long)q * ldp + p];
    const int len = list_len[L];
    const int start = s * seg_codes;
    if (start >= len) continue;
    const int end = min(len, start + seg_codes);

    const float* __restrict__ qv = Qp + (long)q * ld;
    for (int i = threadIdx.x; i < dim; i += ADC_THREADS) {
        if constexpr (HAS_CENTROID) res[i] = qv[i] - centroids[(long)L * ld + i];   // queryResidual[d] = q[d] - centroid[d]
        else res[i] = qv[i];
    }
    __syncthreads();
    const int n_ent = M * KL;
    if constexpr (DSUB > 0) {
        // table build, specialised on the subspace width. The codebook is L2-resident and re-read by every workgroup
        // (786 KiB per (query, list) at M=96), so the loop is software-pipelined: the codebook rows of the NEXT group
        // of U entries are already in flight (16-byte loads) while the current group is reduced. The per-entry
        // float32 sum keeps the reference's dimension order (ivfpq_index_search.go:365-371).
        constexpr int U = (DSUB <= 4) ? 8 : (DSUB <= 8 ? 4 : 2);
        // entry e = m * KL + k with KL a power of two dividing the workgroup size: a thread keeps one codeword index k
        // and walks the subspaces m = m0, m0 + mstep, ... with constant pointer strides (no per-entry address math)
        const int kl_shift = 31 - __builtin_clz((unsigned)KL);
        const int k = threadIdx.x & (KL - 1), m0 = threadIdx.x >> kl_shift, mstep = ADC_THREADS >> kl_shift;
        const float* __restrict__ cbp = codebooks + ((long)m0 * Ksub + k) * DSUB;
        const long cstride = (long)mstep * Ksub * DSUB;
        const int n_j = (M - m0 + mstep - 1) / mstep;       // entries owned by this thread (m0 < M iff n_j > 0)
        float cur[U][DSUB], nxt[U][DSUB];
        auto fetch = [&](float (&dst)[U][DSUB], int j0) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                int j = j0 + u; if (j > n_j - 1) j = n_j - 1; if (j < 0) j = 0;      // clamped re-read, never stored
                const float* __restrict__ cb = cbp + (long)j * cstride;
                if constexpr (DSUB % 4 == 0) {
#pragma unroll
                    for (int i = 0; i < DSUB; i += 4) { const f32x4q v = *reinterpret_cast<const f32x4q*>(cb + i); dst[u][i] = v[0]; dst[u][i + 1] = v[1]; dst[u][i + 2] = v[2]; dst[u][i + 3] = v[3]; }
                } else {
#pragma unroll
                    for (int i = 0; i < DSUB; i++) dst[u][i] = cb[i];
                }
            }
        };
        if (m0 < M) {
            fetch(cur, 0);
            for (int j0 = 0; j0 < n_j; j0 += U) {
                fetch(nxt, j0 + U);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int j = j0 + u;
                    if (j < n_j) {
                        const int m = m0 + j * mstep;
                        const float* r = res + m * DSUB;
                        float dsum = 0.0f;
#pragma unroll
                        for (int i = 0; i < DSUB; i++) { float diff = r[i] - cur[u][i]; float sq = diff * diff; dsum = dsum + sq; }
                        lut[m * KL + k] = dsum;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int i = 0; i < DSUB; i++) cur[u][i] = nxt[u][i];
            }
        }
    } else {
        for (int e = threadIdx.x; e < n_ent; e += ADC_THREADS) {
            const int m = e / KL, k = e - m * KL;
            const float* __restrict__ cb = codebooks + ((long)m * Ksub + k) * dsub;
            const float* r = res + m * dsub;
            float dist = 0.0f;
            for (int i = 0; i < dsub; i++) { float diff = r[i] - cb[i]; float sq = diff * diff; dist = dist + sq; }
            lut[e] = dist;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long base_slot = list_base[L];            // multiple of 64
    // scan: a lane owns one code of each of TWO 64-code blocks (two independent serial float32 chains in flight);
    // the M4 code words are fetched eight at a time, the next group requested before the current one is consumed,
    // and the 32 table lookups of a group are issued as straight-line code so the LDS reads overlap the add chain.
    constexpr int G = 4;
    const int nblk_end = (end + 63) >> 6;
    for (int blk = (start >> 6) + 2 * wid; blk < nblk_end; blk += 2 * (ADC_THREADS / 64)) {
        const bool has1 = (blk + 1) < nblk_end;
        const unsigned* __restrict__ cw0 = codes + ((base_slot >> 6) + blk) * (long)M4 * 64 + lane;
        const unsigned* __restrict__ cw1 = has1 ? cw0 + (long)M4 * 64 : cw0;
        float acc0 = 0.0f, acc1 = 0.0f;
        unsigned cur0[G], cur1[G], nxt0[G], nxt1[G];
#pragma unroll
        for (int i = 0; i < G; i++) { cur0[i] = (i < M4) ? cw0[(long)i * 64] : 0u; cur1[i] = (i < M4) ? cw1[(long)i * 64] : 0u; }
        for (int w0 = 0; w0 < M4; w0 += G) {
#pragma unroll
            for (int i = 0; i < G; i++) {
                const bool in = (w0 + G + i) < M4;
                nxt0[i] = in ? cw0[(long)(w0 + G + i) * 64] : 0u;
                nxt1[i] = in ? cw1[(long)(w0 + G + i) * 64] : 0u;
            }
#pragma unroll
            for (int i = 0; i < G; i++) {
                const int m = (w0 + i) * 4;
                if (m + 3 < M) {            // wave-uniform
                    const unsigned wa = cur0[i], wb = cur1[i];
                    const float* l0 = lut + (long)m * KL;
                    const float a0 = l0[wa & 255u], a1 = l0[KL + ((wa >> 8) & 255u)], a2 = l0[2 * KL + ((wa >> 16) & 255u)], a3 = l0[3 * KL + (wa >> 24)];
                    const float b0 = l0[wb & 255u], b1 = l0[KL + ((wb >> 8) & 255u)], b2 = l0[2 * KL + ((wb >> 16) & 255u)], b3 = l0[3 * KL + (wb >> 24)];
                    acc0 = acc0 + a0; acc1 = acc1 + b0;
                    acc0 = acc0 + a1; acc1 = acc1 + b1;
                    acc0 = acc0 + a2; acc1 = acc1 + b2;
                    acc0 = acc0 + a3; acc1 = acc1 + b3;
                } else if (m < M) {         // last, partial word (M not a multiple of 4)
                    const unsigned wa = cur0[i], wb = cur1[i];
                    for (int bb = 0; m + bb < M; bb++) {
                        acc0 = acc0 + lut[(m + bb) * KL + ((wa >> (8 * bb)) & 255u)];
                        acc1 = acc1 + lut[(m + bb) * KL + ((wb >> (8 * bb)) & 255u)];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < G; i++) { cur0[i] = nxt0[i]; cur1[i] = nxt1[i]; }
        }
        const int j0 = blk * 64 + lane, j1 = j0 + 64;
        if (j0 >= start && j0 < end) {
            const bool ok = elig ? (elig[base_slot + j0] != 0) : true;
            D[(long)q * ldD + so + j0] = ok ? go_sqrt32q(acc0) : __uint_as_float(EXCLUDED_BITS);
        }
        if (has1 && j1 >= start && j1 < end) {
            const bool ok = elig ? (elig[base_slot + j1] != 0) : true;
            D[(long)q * ldD + so + j1] = ok ? go_sqrt32q(acc1) : __uint_as_float(EXCLUDED_BITS);
        }
    }
  }   // item loop
}
size_t adc_lds_bytes(int M, int Ksub, int dim) { int KL = Ksub < 256 ? Ksub : 256; return ((size_t)M * KL + dim) * sizeof(float); }
void launch_adc_scan(Ctx* c, const float* Qp, int ld, int dim, const float* centroids, const float* codebooks, int M, int Ksub, int dsub,
                     const uint32_t* codes, int M4, const int64_t* list_base, const int32_t* list_len, const uint32_t* probe_list, int ldp,
                     int np, const int32_t* seg_off, const uint8_t* elig, int B, int max_list_len, float* D, int64_t ldD) {
    if (B <= 0 || np <= 0 || max_list_len <= 0) return;
    const int KL = Ksub < 256 ? Ksub : 256;
    const size_t lds = adc_lds_bytes(M, Ksub, dim);
    if (lds > 160 * 1024) COMET_FAIL(COMET_ERR_UNSUPPORTED, "PQ lookup table (%zu bytes) exceeds the 160 KiB LDS of a gfx950 CU", lds);
    const int seg_codes = 8192;
    const int segs = (int)ceil_div(max_list_len, seg_codes);
    const long n_items = (long)B * np * segs;
    dim3 grid((unsigned)std::min<long>(n_items, (long)c->prop.multiProcessorCount)), blk(ADC_THREADS);
    ProfScope ps(c, "adc_scan");
#define ADC_LAUNCH(HC, DS) do { \
        HIP_CHECK(hipFuncSetAttribute((const void*)adc_scan_kernel<HC, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        adc_scan_kernel<HC, DS><<<grid, blk, lds, c->stream>>>(Qp, ld, dim, centroids, codebooks, M, Ksub, KL, dsub, codes, M4, (const long*)list_base, \
                                                              list_len, probe_list, ldp, np, seg_off, elig, D, ldD, segs, seg_codes, B); } while (0)
#define ADC_DS(HC) do { switch (dsub) { case 2: ADC_LAUNCH(HC, 2); break; case 4: ADC_LAUNCH(HC, 4); break; case 8: ADC_LAUNCH(HC, 8); break; \
                                       case 16: ADC_LAUNCH(HC, 16); break; default: ADC_LAUNCH(HC, 0); break; } } while (0)
    if (centroids) ADC_DS(true); else ADC_DS(false);
#undef ADC_DS
#undef ADC_LAUNCH
    LAUNCH_CHECK();
}

}  // namespace comet
