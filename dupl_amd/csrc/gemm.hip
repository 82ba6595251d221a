// Exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why f32-input MFMA: the parity bar is CAM max-abs-diff < 1e-3 *and identical label
// maps*; the f32 MFMA is bit-for-bit an fmaf chain, runs at the f32 vector peak (157 TF/s) and
// leaves the VALU free for the fused epilogues.  The kernel is MFMA-bound by construction: per
// k-step of 2 a wave issues MI x NJ MFMAs (64 cycles each) against MI + NJ ds_read_b32.
//
// Block = 256 threads = 4 waves (2 x 2).  Tile instantiations (BM x 64*NJ x 32):
//   64 x 128  default: wave tile 32 x 64 (1 x 2 MFMA tiles), 4 resident blocks / CU;
//   64 x  64  small grids and every split-K weight gradient: wave tile 32 x 32, twice the blocks;
//  128 x 128  wave tile 64 x 64 (2 x 2), kept for the tuning knob (measured slower on the DuPL shapes).
// Both operand tiles live in LDS k-major ([k][m], [k][n]) so the MFMA fragment reads (lane -> consecutive m / n) are
// bank-conflict free; the global->LDS stage goes through registers (prefetch of tile t+1 overlaps the MFMAs of tile t)
// and transposes on the LDS write when the operand is k-contiguous in memory (odd row stride: conflict-free scatter).
// Grid: one block per output tile; XCD x (private L2) owns a contiguous band of tiles, ordered inside the band in groups
// of 16 row-tiles x all column tiles (row-tile fastest); blockIdx.y = batch, blockIdx.z = k-split of pure accumulate
// GEMMs (weight gradients), combined with f32 atomics.
#include <type_traits>

#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

constexpr int BN = 128, BK = 32, NT = 256;

// One float4 chunk c of a (ROWS x 32) operand tile, fully guarded (edge tiles, unaligned rows, k tails).
// KC (k-contiguous): element (r, k) at base[r*ld + k]; chunk c -> row c>>3, k (c&7)*4.
// MC (m-contiguous): element (r, k) at base[k*ld + r]; chunk c -> k c/(ROWS/4), r (c%(ROWS/4))*4.
template <bool MC, int ROWS>
__device__ __forceinline__ float4 load_chunk(const float* __restrict__ base, int ld, int r0, int k0, int R, int K, bool vec_ok,
                                             int c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!MC) {
        const int r = r0 + (c >> 3), k = k0 + ((c & 7) << 2);
        if (r < R && k < K) {
            const float* p = base + (size_t)r * ld + k;
            if (vec_ok && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
            else {
                v.x = p[0];
                if (k + 1 < K) v.y = p[1];
                if (k + 2 < K) v.z = p[2];
                if (k + 3 < K) v.w = p[3];
            }
        }
    } else {
        const int k = k0 + c / (ROWS / 4), r = r0 + ((c % (ROWS / 4)) << 2);
        if (k < K && r < R) {
            const float* p = base + (size_t)k * ld + r;
            if (vec_ok && r + 3 < R) v = *reinterpret_cast<const float4*>(p);
            else {
                v.x = p[0];
                if (r + 1 < R) v.y = p[1];
                if (r + 2 < R) v.z = p[2];
                if (r + 3 < R) v.w = p[3];
            }
        }
    }
    return v;
}

// LDS tiles are k-major ([k][row], row stride S): KC chunks are transposed on the way in (4 scalar stores, S odd ->
// conflict free), MC chunks go in as one 16-byte store.
template <bool MC, int S, int ROWS>
__device__ __forceinline__ void store_chunk(const float4 v, float* __restrict__ lds, int c) {
    if (!MC) {
        const int r = c >> 3, k = (c & 7) << 2;
        lds[(k + 0) * S + r] = v.x;
        lds[(k + 1) * S + r] = v.y;
        lds[(k + 2) * S + r] = v.z;
        lds[(k + 3) * S + r] = v.w;
    } else {
        const int k = c / (ROWS / 4), r = (c % (ROWS / 4)) << 2;
        *reinterpret_cast<float4*>(&lds[k * S + r]) = v;
    }
}

// BM = 128: wave tile 64x64 (2x2 MFMA tiles).  BM = 64: wave tile 32x64 (1x2) -- twice the blocks for small grids.
// NJ = 32-column MFMA tiles per wave along n: 2 -> 128-column block tile (default), 1 -> 64-column tile (twice the
// blocks for N = 768 outputs on small-M grids: the data-gradient GEMMs of the training batch).
template <bool A_MC, bool B_NC, int BM, int FAST, int NJ = 2>   // FAST: 0 guarded loader, 1 branch-free, 2 + k-range mask
__global__ __launch_bounds__(NT) void gemm_f32_kernel(const dupl_gemm_desc p, const int g_gm) {
    constexpr int BN = 64 * NJ;                 // shadows the 128-column default
    constexpr int MI = BM / 64;                 // 32-row MFMA tiles per wave along m
    constexpr int SA = A_MC ? (BM + 4) : (BM + 1);
    constexpr int SB = B_NC ? (BN + 4) : (BN + 1);
    __shared__ __attribute__((aligned(16))) float smem[BK * SA + BK * SB];
    float* As = smem;
    float* Bs = smem + BK * SA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hf = lane >> 5;

    // ---- tile id with XCD-aware (bijective) remap: XCD x = bid % 8 owns a contiguous band
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nblk = nbm * nbn;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // grouped order inside the band: GM row-tiles x all n-tiles at a time, row-tile fastest, so that the ~128 blocks
    // resident on one XCD cover a GM x (128/GM) patch of C (A and B panels of similar size) instead of 5 row-tiles x
    // every n-tile (which re-streams the whole weight matrix through the 4 MB L2 for every 5 row-tiles)
    const int gspan = g_gm * nbn;
    const int gid = lid / gspan, gin = lid - gid * gspan;
    const int gfirst = gid * g_gm;
    const int gsz = min(nbm - gfirst, g_gm);
    const int tm = gfirst + gin % gsz, tn = gin / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.y;
    const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
    const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const float* B = p.B + z0 * p.sB0 + z1 * p.sB1;
    float* C = p.C + z0 * p.sC0 + z1 * p.sC1;

    const bool a_vec = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // split-K (gridDim.z > 1): this block reduces k in [kbeg, kend) and adds its partial tile atomically
    const int ksplit = gridDim.z;
    int kbeg = 0, kend = p.K;
    if (ksplit > 1) {
        const int kchunk = ((p.K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
        kbeg = blockIdx.z * kchunk;
        kend = min(p.K, kbeg + kchunk);
        if (kbeg >= kend) return;
    }
    const int nt = (kend - kbeg + BK - 1) / BK;
    // FAST (decided on the host, dupl_gemm_f32): every row 16-byte aligned (and K % 4 == 0 for k-contiguous
    // operands) -> branch-free clamped float4 loads with a multiplicative k-range mask; otherwise the guarded loader.
    const float* a_rd = As + wm * (BM / 2) + l31;
    const float* b_rd = Bs + wn * (32 * NJ) + l31;

    // ---- k-loop.  Staging registers are plain local float4 arrays filled by fully inlined code (no struct refs).
    constexpr int NA = BM / 32, NB = BN / 32;   // float4 per thread per k-tile (256 threads)
    float4 ra[NA], rb[NB];
    auto issue_loads = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + NT * i;
            if (FAST) {
                // branch-free: clamp the address into the matrix, then zero the chunk if its k lies past the range
                const int kk = A_MC ? k0 + c / (BM / 4) : k0 + ((c & 7) << 2);
                const int kc = min(kk, A_MC ? kend - 1 : kend - 4);
                float4 v;
                if (!A_MC) v = *reinterpret_cast<const float4*>(A + (size_t)min(m0 + (c >> 3), p.M - 1) * p.lda + kc);
                else v = *reinterpret_cast<const float4*>(A + (size_t)kc * p.lda + min(m0 + ((c % (BM / 4)) << 2), p.M - 4));
                const bool keep = FAST == 1 || kk < kend;   // FAST == 1: K is a multiple of the k-tile, no mask needed
                ra[i] = make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
            } else {
                ra[i] = load_chunk<A_MC, BM>(A, p.lda, m0, k0, p.M, kend, a_vec, c);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = tid + NT * i;
            if (FAST) {
                const int kk = B_NC ? k0 + c / (BN / 4) : k0 + ((c & 7) << 2);
                const int kc = min(kk, B_NC ? kend - 1 : kend - 4);
                float4 v;
                if (!B_NC) v = *reinterpret_cast<const float4*>(B + (size_t)min(n0 + (c >> 3), p.N - 1) * p.ldb + kc);
                else v = *reinterpret_cast<const float4*>(B + (size_t)kc * p.ldb + min(n0 + ((c % (BN / 4)) << 2), p.N - 4));
                const bool keep = FAST == 1 || kk < kend;
                rb[i] = make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
            } else {
                rb[i] = load_chunk<B_NC, BN>(B, p.ldb, n0, k0, p.N, kend, b_vec, c);
            }
        }
    };
    issue_loads(kbeg);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) store_chunk<A_MC, SA, BM>(ra[i], As, tid + NT * i);
#pragma unroll
        for (int i = 0; i < NB; ++i) store_chunk<B_NC, SB, BN>(rb[i], Bs, tid + NT * i);
        __syncthreads();
        if (t + 1 < nt) issue_loads(kbeg + (t + 1) * BK);
        // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
        float fa[2][MI], fb[2][NJ];
        {
            const int k = hf;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[0][i] = a_rd[k * SA + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[0][j] = b_rd[k * SB + 32 * j];
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < BK / 2) {
                const int k = 2 * (s + 1) + hf;
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[nxt][i] = a_rd[k * SA + 32 * i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[nxt][j] = b_rd[k * SB + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the next step's ds_reads ahead of this step's MFMAs
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const float* bias = p.bias ? p.bias + z0 * p.sBias0 + z1 * p.sBias1 : nullptr;
    const float* res = p.res ? p.res + z0 * p.sR0 + z1 * p.sR1 : nullptr;
    const float* aux = p.aux ? p.aux + z0 * p.sX0 + z1 * p.sX1 : nullptr;
    const int fl = p.flags;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * (32 * NJ) + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hf;
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][e] + bv;
                if (fl & DUPL_GEMM_STORE_PRE) const_cast<float*>(aux)[(size_t)row * p.ldaux + col] = v;
                if (fl & DUPL_GEMM_GELU) v = gelu_f(v);
                if (fl & DUPL_GEMM_RELU) v = fmaxf(v, 0.f);
                if (fl & DUPL_GEMM_ABS) v = fabsf(v);
                if (fl & DUPL_GEMM_MUL_DGELU) v *= gelu_grad_f(aux[(size_t)row * p.ldaux + col]);
                if (fl & DUPL_GEMM_MUL_RELUMASK) v = aux[(size_t)row * p.ldaux + col] > 0.f ? v : 0.f;
                if (res) v += res[(size_t)row * p.ldr + col];
                float* cp = C + (size_t)row * p.ldc + col;
                if (ksplit > 1) { unsafeAtomicAdd(cp, v); continue; }   // hardware global_atomic_add_f32
                if (fl & DUPL_GEMM_ACCUM) v += *cp;
                *cp = v;
            }
        }
    }
}

}  // namespace

extern "C" int dupl_gemm_f32(const dupl_gemm_desc* d, dupl_stream_t stream) {
    if (!d || d->struct_size != sizeof(dupl_gemm_desc)) return DUPL_ERR_ARG;       // a caller built against another header
    if (!d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->zdiv <= 0)
        return DUPL_ERR_ARG;
    // launch tuning travels in the descriptor (0 = heuristic): tile_rows 64 / 128, tile_cols 64 / 128, group 1 .. 4096
    if ((d->tile_rows != 0 && d->tile_rows != 64 && d->tile_rows != 128) || (d->tile_cols != 0 && d->tile_cols != 64 && d->tile_cols != 128) ||
        d->group < 0 || d->group > 4096)
        return DUPL_ERR_ARG;
    const int g_tile_override = d->tile_rows, g_ncols_override = d->tile_cols, g_group_m = d->group ? d->group : 16;
    if ((d->flags & (DUPL_GEMM_MUL_DGELU | DUPL_GEMM_MUL_RELUMASK | DUPL_GEMM_STORE_PRE)) && !d->aux) return DUPL_ERR_ARG;
    const bool amc_ = d->flags & DUPL_GEMM_A_MCONTIG, bnc_ = d->flags & DUPL_GEMM_B_NCONTIG;
    // 64-row tiles (4 resident blocks / CU) measured >= 128-row tiles on every DuPL shape (profiles/r01_gemm_tiles.txt)
    bool small = true;
    if (g_tile_override) small = g_tile_override == 64;
    const int bm = small ? 64 : 128;
    const int nbm = (d->M + bm - 1) / bm;
    const int pure = DUPL_GEMM_A_MCONTIG | DUPL_GEMM_B_NCONTIG | DUPL_GEMM_ACCUM;
    const bool splitk_ok = (d->flags & ~pure) == 0 && (d->flags & DUPL_GEMM_ACCUM) && !d->bias && !d->res && d->alpha == 1.0f;
    // 64-column tiles when 64 x 128 tiles leave most of the 1024 block slots empty (the N = 768 data gradients of a
    // 3 140-token batch: 49 x 6 = 294 blocks -> 588) and for every split-K weight gradient (half the k-splits, i.e.
    // half the atomics, for the same number of blocks: +8..35 % measured)
    int ncols = 128;
    if (small && (splitk_ok || (long)nbm * ((d->N + 127) / 128) * d->batch < 640)) ncols = 64;
    if (g_ncols_override && small) ncols = g_ncols_override;
    const bool n64 = ncols == 64;
    const int nbn = (d->N + ncols - 1) / ncols;
    // split-K for pure accumulate GEMMs (weight gradients: tiny M x N, K = all tokens): fill >= ~4 blocks per CU
    int ksplit = 1;
    if (splitk_ok) {
        const long blocks = (long)nbm * nbn * d->batch;
        if (blocks < 1024) {
            ksplit = (int)((1024 + blocks - 1) / blocks);
            const int maxs = (d->K + 127) / 128;   // keep >= 128 k per split
            if (ksplit > maxs) ksplit = maxs;
            if (ksplit < 1) ksplit = 1;
        }
    }
    if (d->deterministic) ksplit = 1;          // no fp32 atomics: one block owns the whole reduction of its tile
    dim3 grid(nbm * nbn, d->batch, ksplit), block(NT);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool amc = d->flags & DUPL_GEMM_A_MCONTIG, bnc = d->flags & DUPL_GEMM_B_NCONTIG;
    // fast-path predicate (see gemm_mainloop): 16-byte aligned operands incl. batch strides, K a multiple of the
    // k-tile (split-K chunks are), m-/n-contiguous operands with a row count that is a multiple of 4
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool fast = al16(d->A) && al16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) && d->K >= 4 &&
                (d->sA0 % 4 == 0) && (d->sA1 % 4 == 0) && (d->sB0 % 4 == 0) && (d->sB1 % 4 == 0);
    if (amc_) fast = fast && (d->M % 4 == 0) && d->M >= 4;
    else fast = fast && (d->K % 4 == 0);          // float4 chunks run along k: whole chunks are in or out
    if (bnc_) fast = fast && (d->N % 4 == 0) && d->N >= 4;
    else fast = fast && (d->K % 4 == 0);
    const bool kfull = (d->K % BK) == 0;   // split-K chunks are multiples of BK, so only the global tail matters
#define DUPL_GEMM_LAUNCH(AM, BNC)                                                                                  \
    do {                                                                                                           \
        if (n64 && fast && kfull) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 64, 1, 1>), grid, block, 0, s, *d, g_group_m); \
        else if (n64 && fast) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 64, 2, 1>), grid, block, 0, s, *d, g_group_m);     \
        else if (n64) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 64, 0, 1>), grid, block, 0, s, *d, g_group_m);             \
        else if (small && fast && kfull) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 64, 1>), grid, block, 0, s, *d, g_group_m);  \
        else if (small && fast) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 64, 2>), grid, block, 0, s, *d, g_group_m);      \
        else if (small) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 64, 0>), grid, block, 0, s, *d, g_group_m);              \
        else if (fast && kfull) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 128, 1>), grid, block, 0, s, *d, g_group_m);     \
        else if (fast) DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 128, 2>), grid, block, 0, s, *d, g_group_m);              \
        else DUPL_LAUNCH((gemm_f32_kernel<AM, BNC, 128, 0>), grid, block, 0, s, *d, g_group_m);                        \
    } while (0)
    if (!amc && !bnc) DUPL_GEMM_LAUNCH(false, false);
    else if (!amc && bnc) DUPL_GEMM_LAUNCH(false, true);
    else if (amc && !bnc) DUPL_GEMM_LAUNCH(true, false);
    else DUPL_GEMM_LAUNCH(true, true);
#undef DUPL_GEMM_LAUNCH
    return dupl_launch_status();
}

extern "C" int dupl_abi_version(void) { return 4; }
