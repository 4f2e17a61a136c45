// K5w-t: weight / bias gradient of a stacked Linear pair for WIDE layers on LARGE graphs (hidden >= 256, BASELINE
// config 5):  dW[o,i] (+)= sum_n G[n,o] * X[n,i],  db[o] (+)= sum_n G[n,o]   (autograd backward of nn.Linear at
// reference impl/models.py:158-159, 169-170).
//
// linear.hip's kernel feeds the MFMAs straight from global memory with a 128 x 64 output tile per workgroup; at
// O = I = 512 every G row is then re-read by 8 input tiles and every X row by 4 output tiles — 24.6 GB per launch at
// N = 1 M, HBM-bound at 5.1 ms for 3.5 ms of matrix work (profiles/r02: 83 - 103 TF/s).  Here both operands go through
// LDS: workgroup tile 128 outputs x 256 inputs (4 waves as 2 x 2, 128 accumulator registers each), 16 rows of the node
// dimension per stage, double buffered: 12 GB per launch, and a stage's 64 MFMAs (4 096 cycles per wave) cover its
// loads.  The LDS images are the plain row-major [16 rows][128 | 256] tiles (the node index IS the MFMA k index, so
// staging is a straight 16-B copy and an MFMA operand is one ds_read_b32, consecutive lanes consecutive words).
// Split over row slabs; slab partials are plain [128][256] tiles summed in slab order by the reduce kernel
// (deterministic).  fp32 MFMA = exact k-ordered fmaf chain.
#include "wgrad_common.h"
#include "dense_common.h"
#include "split_mma.h"

namespace glass {



constexpr int kWO = 128, kWI = 256, kWK = 16, kWThreads = 256;
constexpr int kWStage = kWK * (kWO + kWI);  // floats per stage: 6 144 = 24 KiB
constexpr int kWTile = kWO * kWI;
// Split products (split_mma.h): a stage holds the two operands as bf16 pieces TRANSPOSED — the MFMA's k is the node row, so a
// fragment is 8 consecutive node rows of one column: [piece][h][slot] 16-byte units.  The staging thread therefore takes
// consecutive node rows of its column quad (2 of the A tile, 4 of the B tile) and writes one packed word / pair per column
// and piece; slot <-> column is a permutation chosen so that those writes run over consecutive units: A slot 32 e + q <->
// output 4 q + e, B slot 128 (q >> 5) + 32 e + (q & 31) <-> input 4 q + e (q = the thread's column quad, e = 0..3).
constexpr int kWStageS = 4 * (SplitImg<kWO>::kUnits + SplitImg<kWI>::kUnits);  // floats per stage: 9 216 = 36 KiB
// Mode header the partial kernel leaves behind the bias partials for the reduce kernel (which is launched later, by
// glass_linear_wgrad_reduce_batch_f32, from (N, O, I) alone): [0] = 1.0f when the partials are in effective-weight form,
// [1] = z_ratio.
constexpr int kWHeaderFloats = 4;

bool wgrad_tiled_shape(int64_t N, int64_t O, int64_t I) { return O >= 512 && O % kWO == 0 && I % kWI == 0 && N >= 65536; }

TiledWgradGeom wgrad_tiled_geom(int64_t N, int64_t O, int64_t I) {
    TiledWgradGeom g;
    g.ny = (int)(I / kWI);
    g.nz = (int)(O / kWO);
    // about two resident workgroups per CU over all (slab, tile) pairs, slabs of whole 16-row stages, at most 256 slabs
    int64_t slabs = ceil_div(512, (int64_t)g.ny * g.nz);
    if (I == O) slabs *= 2;  // comb-shaped: in the effective-weight mode half of the output tiles hold labeled rows only
    if (slabs > 256) slabs = 256;
    int64_t rows = ceil_div(ceil_div(N, slabs), kWK) * kWK;
    g.rows_per_slab = (int)rows;
    g.n_slabs = (int)ceil_div(N, rows);
    g.part_w_floats = (int64_t)g.n_slabs * g.ny * g.nz * kWTile;
    g.part_b_floats = (int64_t)g.n_slabs * g.nz * kWO + kWHeaderFloats;  // + the mode header behind the bias partials
    return g;
}

// One stage of raw loads in registers (issued a whole stage of MFMAs before they are needed) and their commit to LDS.
struct StageCtx {
    const float* G; int64_t ldg;
    const float* xsrc; int64_t xld; int xcol;
    int a_col; bool a_first;
    int64_t r0, r1;
    int tid;
    bool eff, lab;  // effective-weight mode; labeled-rows tile of it
};

template <bool SYNTH, bool S3>
struct WStageRegs {
    float4 ga[2], ta[2], xb[4];
    int mk[2];
    bool alive[2];
    __device__ __forceinline__ void issue(const StageCtx& c, const WgradSynth& sy, int step) {
        const int64_t base = c.r0 + (int64_t)step * kWK;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int64_t n = base + (S3 ? 2 * (c.tid >> 5) + a : (c.tid + kWThreads * a) >> 5);
            alive[a] = n < c.r1;
            const int64_t nn = alive[a] ? n : c.r1 - 1;  // clamped: loads never wait on a predicate, zeroed at use
            if (!SYNTH) {
                ga[a] = *reinterpret_cast<const float4*>(c.G + nn * c.ldg + c.a_col);
            } else {
                ga[a] = *reinterpret_cast<const float4*>(sy.dsrc + nn * sy.ldd + (c.a_first ? c.a_col : c.a_col - sy.H));
                if (sy.act != GLASS_ACT_NONE) ta[a] = *reinterpret_cast<const float4*>(sy.T + nn * sy.ldt + c.a_col);
                mk[a] = sy.mask[nn];
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t n = base + (S3 ? 4 * (c.tid >> 6) + b : (c.tid + kWThreads * b) >> 6);
            const int64_t nn = n < c.r1 ? n : c.r1 - 1;  // rows past the slab: the A operand is zero there
            xb[b] = *reinterpret_cast<const float4*>(c.xsrc + nn * c.xld + c.xcol);
        }
    }
    __device__ __forceinline__ void commit(const StageCtx& c, const WgradSynth& sy, float* stage, float4& bsum) const {
        float4 gv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float4 g = ga[a];
            if (SYNTH) {
                const float cf = c.eff ? ((c.lab && mk[a] == 0) ? 0.f : 1.f) : (((mk[a] != 0) == c.a_first) ? sy.zr : sy.omz);
                g.x *= cf; g.y *= cf; g.z *= cf; g.w *= cf;
                if (sy.act != GLASS_ACT_NONE) {
                    g.x *= act_grad(sy.act, ta[a].x); g.y *= act_grad(sy.act, ta[a].y); g.z *= act_grad(sy.act, ta[a].z); g.w *= act_grad(sy.act, ta[a].w);
                }
            }
            if (!alive[a]) g = make_float4(0.f, 0.f, 0.f, 0.f);
            bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
            if (S3) gv[a] = g;
            else reinterpret_cast<float4*>(stage)[c.tid + kWThreads * a] = g;
        }
        if (S3) {
            // A: node rows 2 g2, 2 g2 + 1 (g2 = tid >> 5) of column quad qa = tid & 31: one packed pair per column and piece
            const int qa = c.tid & 31, g2 = c.tid >> 5;
            unsigned* A = reinterpret_cast<unsigned*>(stage);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hi, mid, lo;
                split2(f4e(gv[0], e), f4e(gv[1], e), hi, mid, lo);
                const int u = (((g2 >> 2) * kWO + e * 32 + qa) << 2) + (g2 & 3);
                A[u] = hi;
                A[4 * SplitImg<kWO>::kPlane + u] = mid;
                A[8 * SplitImg<kWO>::kPlane + u] = lo;
            }
            // B: node rows 4 g .. 4 g + 3 (g = tid >> 6) of column quad q = tid & 63: 8 bytes per column and piece
            const int q = c.tid & 63, g = c.tid >> 6;
            uint2* B = reinterpret_cast<uint2*>(stage + 4 * SplitImg<kWO>::kUnits);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint2 hi, mid, lo;
                split2(f4e(xb[0], e), f4e(xb[1], e), hi.x, mid.x, lo.x);
                split2(f4e(xb[2], e), f4e(xb[3], e), hi.y, mid.y, lo.y);
                const int u = (((g >> 1) * kWI + (q >> 5) * 128 + e * 32 + (q & 31)) << 1) + (g & 1);
                B[u] = hi;
                B[2 * SplitImg<kWI>::kPlane + u] = mid;
                B[4 * SplitImg<kWI>::kPlane + u] = lo;
            }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) reinterpret_cast<float4*>(stage + kWK * kWO)[c.tid + kWThreads * b] = xb[b];
        }
    }
};

// EFF (comb pair: no activation factor in G): G[n, o] is dc[n, o mod H] times a coefficient that depends only on the row's
// label and on the half o belongs to, so the f1 and f0 halves of dW are two scalings of the SAME matrix for every unlabeled
// row.  The lower half of the output tiles then accumulates S = sum over ALL rows of dc^T X (coefficient 1), the upper half
// L = the same sum over the LABELED rows only — stages of 16 rows without a labeled row are skipped there, and with
// B*Smax labeled nodes among N almost all are — and the reduce kernel forms  dW1 = (1-z) S + (2z-1) L,  dW0 = z S - (2z-1) L
// (bias likewise).  Half the matrix work of the two-product form.
template <bool SYNTH, bool EFF, bool S3>
__global__ __launch_bounds__(kWThreads, 2) void tiled_wgrad_kernel(const float* __restrict__ G, int64_t ldg,
                                                                  const float* __restrict__ X, int64_t ldx, int64_t N,
                                                                  int rows_per_slab, float* __restrict__ part_w,
                                                                  float* __restrict__ part_b, float* __restrict__ header,
                                                                  WgradSynth sy, int z0, int nz_all) {
    // (z0, nz_all): this launch covers output tiles z0 .. z0 + gridDim.z - 1 of nz_all — the split form launches the
    // all-rows tiles on tiled_wgrad8_kernel and only the labeled-rows tiles here
    extern __shared__ float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1, j = lane & 31, h = lane >> 5;
    const int bz = blockIdx.z + z0;
    if (header && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
        header[0] = EFF ? 1.f : 0.f;
        header[1] = sy.zr;
        header[2] = 0.f;
    }
    const bool lab_tile = EFF && bz >= nz_all / 2;  // the labeled-rows sum of output tile z - nz/2
    const int o0 = (lab_tile ? bz - nz_all / 2 : bz) * kWO, i0 = blockIdx.y * kWI;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab;
    const int64_t r1 = min(N, r0 + rows_per_slab);
    const int n_steps = (int)((r1 - r0 + kWK - 1) / kWK);
    // staging assignment.  A tile [16][128]: float4 f = tid + 256 a: row f >> 5, column quad f & 31 (a = 0, 1);
    // B tile [16][256]: float4 f = tid + 256 b: row f >> 6, column quad f & 63 (b = 0..3).  The column quads are the
    // same for every stage, so the bias partial of this thread's four outputs accumulates in registers.
    const int a_col = o0 + 4 * (tid & 31);          // global output column of this thread's A quad
    const bool a_first = a_col < sy.H;              // SYNTH: f1 half
    const int b_col = i0 + 4 * (tid & 63);
    const float* xsrc = X;
    int64_t xld = ldx;
    int xcol = b_col;
    if (SYNTH && sy.X2 != nullptr && b_col >= sy.H) {  // second half of the virtual concatenation [X | X2]
        xsrc = sy.X2;
        xld = sy.ldx2;
        xcol = b_col - sy.H;
    }
    constexpr int kStage = S3 ? kWStageS : kWStage;  // floats per stage
    WStageRegs<SYNTH, S3> sr;
    const StageCtx cx{G, ldg, xsrc, xld, xcol, a_col, a_first, r0, r1, tid, EFF, lab_tile};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[rb][cb][k] = 0.f;

    auto stage_mma = [&](const float* cur) __attribute__((always_inline)) {
        if constexpr (S3) {
            const float4* Ai = reinterpret_cast<const float4*>(cur);
            const float4* Bi = Ai + SplitImg<kWO>::kUnits;
            uint4 a[2][3];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) SplitImg<kWO>::frag(Ai, wm * 64 + rb * 32 + j, h, a[rb]);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                uint4 b[3];
                SplitImg<kWI>::frag(Bi, wn * 128 + cb * 32 + j, h, b);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) split_mma(acc[rb][cb], a[rb], b);
            }
            return;
        }
        const float* A = cur + wm * 64 + j;
        const float* B = cur + kWK * kWO + wn * 128 + j;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = 8 * h + s;  // node row of this stage fed by this lane half at MFMA step s
            float av[2], bv[4];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) av[rb] = A[k * kWO + rb * 32];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bv[cb] = B[k * kWI + cb * 32];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rb], bv[cb], acc[rb][cb], 0, 0, 0);
        }
    };
    if (lab_tile) {
        // labeled rows only: one flag per 16-row stage first (LDS, behind the two stage buffers), then only the flagged
        // stages are loaded and multiplied — few, so without the software pipeline
        int* flag = reinterpret_cast<int*>(wsm + 2 * kStage);
        for (int st0 = 0; st0 < n_steps; st0 += kWThreads) {
            const int stp = st0 + tid;
            if (stp < n_steps) {
                const int64_t b = r0 + (int64_t)stp * kWK;
                int any = 0;
                if (b + kWK <= r1 && ((reinterpret_cast<uintptr_t>(sy.mask + b) & 15u) == 0)) {
                    const uint4 m4 = *reinterpret_cast<const uint4*>(sy.mask + b);
                    any = (m4.x | m4.y | m4.z | m4.w) != 0;
                } else {
                    for (int64_t n = b; n < b + kWK && n < r1; ++n) any |= sy.mask[n] != 0;
                }
                flag[stp] = any;
            }
        }
        __syncthreads();
        for (int step = 0; step < n_steps; ++step) {
            if (!flag[step]) continue;  // workgroup-uniform
            sr.issue(cx, sy, step);
            sr.commit(cx, sy, wsm, bsum);
            __syncthreads();
            stage_mma(wsm);
            __syncthreads();
        }
    } else {
        sr.issue(cx, sy, 0);
        sr.commit(cx, sy, wsm, bsum);
        __syncthreads();
        for (int step = 0; step < n_steps; ++step) {
            const float* cur = wsm + (step & 1) * kStage;
            float* nxt = wsm + ((step + 1) & 1) * kStage;
            if (step + 1 < n_steps) sr.issue(cx, sy, step + 1);
            stage_mma(cur);
            if (step + 1 < n_steps) sr.commit(cx, sy, nxt, bsum);
            __syncthreads();
        }
    }

    // partial tile, plain [128][256]: acc[rb][cb][k] = output wm*64 + rb*32 + 8(k>>2) + 4h + (k&3), input wn*128 + cb*32 + j
    const int64_t tile_id = ((int64_t)bz * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* pw = part_w + tile_id * kWTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (S3) {  // slot -> column as the staging laid them out: four consecutive inputs per lane
                const int o = 4 * (8 * (k >> 2) + 4 * h + (k & 3)) + 2 * wm + rb;
                *reinterpret_cast<float4*>(pw + o * kWI + 4 * (32 * wn + j)) =
                    make_float4(acc[rb][0][k], acc[rb][1][k], acc[rb][2][k], acc[rb][3][k]);
            } else {
                const int o = wm * 64 + rb * 32 + 8 * (k >> 2) + 4 * h + (k & 3);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) pw[o * kWI + wn * 128 + cb * 32 + j] = acc[rb][cb][k];
            }
        }
    if (blockIdx.y == 0 && part_b) {
        // 8 threads (tid & 31 equal) hold partial bias sums of the same 4 outputs: combine through LDS in thread order
        float4* red = reinterpret_cast<float4*>(wsm);  // the loop's last barrier freed the stages
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float4 s = red[tid];
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                const float4 o = red[tid + 32 * r];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *reinterpret_cast<float4*>(part_b + ((int64_t)bz * gridDim.x + blockIdx.x) * kWO + 4 * tid) = s;
        }
    }
}

// ---- split form, all-rows tiles: eight waves, three stages of raw rows in flight --------------------------------------
// The four-wave kernel above keeps ONE stage of raw loads in registers, issued a single MFMA phase (0.7 us) before it is cut:
// every step waits out the rest of the operand rows' round trip, and two workgroups per CU only half hide it (MFMA busy 0.36 -
// 0.40 at config 5; 249 registers: no room for a second set).  Here the same 128 x 256 tile and the same LDS image belong to
// EIGHT waves — wave (wm, wn) owns 64 outputs x 64 inputs: 64 accumulator registers —, so a lane has room for THREE sets of
// raw rows: stage k is requested three steps before it is cut (under the MFMAs of steps k - 4 .. k - 2), the hand-over
// barrier waits for LDS only, and the two waves of a SIMD fill each other's cut / LDS phases with MFMAs.
//   * No conditional memory instruction in the loop (common.h, lds_barrier): buffer loads on resources that start at the
//     slab's first row and end behind its last — a row past the slab reads as zero, so there is no clamp, no liveness flag
//     and the loop runs whole triples of steps.  Waves 0-3 stage the gradient operand (2 rows x one column quad: gradient,
//     pre-activation, the rows' label bytes as ONE 16-bit load), waves 4-7 the input operand (4 rows x one column quad):
//     the same five load instructions in either kind of wave, on resources selected per wave in scalar registers.
//   * Every use of a set's values is pinned behind the point where its stage is cut (glass_pin) — the compiler otherwise
//     computes a LATER stage's label coefficient or zero-extension early and waits for the youngest loads there; the
//     file is compiled without SLP vectorisation for the same reason (Makefile).
//   * LDS writes without bank conflicts: a wave's lanes write consecutive words — lane l of an A wave holds row pair
//     4 (w & 1) + (l & 3) of column quad (l >> 2) + 16 (w >> 1); lane l of a B wave row quad 2 (w & 1) + (l & 1) of column quad
//     (l >> 1) + 32 (w >> 1) (the four-wave mapping wrote 4-way conflicts: 0.45 of its LDS cycles).
constexpr int kW8Threads = 512;
struct W8Set {
    float4 v[4];          // A wave: gradient rows 0, 1, pre-activation rows 0, 1; B wave: input rows 0 .. 3
    unsigned short mk2;   // A wave: the two rows' label bytes (kept as loaded: the zero-extension is a use)
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int k = 0; k < 4; ++k) { glass_pin(v[k].x); glass_pin(v[k].y); glass_pin(v[k].z); glass_pin(v[k].w); }
        glass_pin(mk2);
    }
};

template <bool EFF>
__global__ __launch_bounds__(kW8Threads, 1) void tiled_wgrad8_kernel(const float* __restrict__ X, int64_t ldx, int64_t N,
                                                                    int rows_per_slab, float* __restrict__ part_w,
                                                                    float* __restrict__ part_b, float* __restrict__ header,
                                                                    WgradSynth sy, int n_slabs, int ny, int nz) {
    extern __shared__ float wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the per-wave resource selection below stays in SGPRs)
    const int wm = w & 1, wn = w >> 1, j = lane & 31, h = lane >> 5;
    if (header && blockIdx.x == 0 && tid == 0) {
        header[0] = EFF ? 1.f : 0.f;
        header[1] = sy.zr;
        header[2] = 0.f;
    }
    // XCD-aware placement (1-D grid; consecutive workgroup ids go round the 8 XCDs, each with its own L2): the ny * nz tiles
    // of ONE slab — which read the same rows: every input tile the slab's gradient rows, every output tile its input rows —
    // get consecutive ids on the SAME XCD (output tile fastest), so they run side by side and the re-reads are L2 hits
    // instead of HBM reads.  With the plain (slab, y, z) grid the z tiles of a slab ran whole rounds apart.
    int bx, by, bz;
    {
        const int lin = blockIdx.x, T = ny * nz;
        int t;
        if (n_slabs % 8 == 0) {
            const int c = lin & 7, m = lin >> 3;
            t = m % T;
            bx = c + 8 * (m / T);
        } else {
            t = lin % T;
            bx = lin / T;
        }
        bz = t % nz;
        by = t / nz;
    }
    const int o0 = bz * kWO, i0 = by * kWI;
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int64_t r1 = min(N, r0 + rows_per_slab);
    const int64_t rows = r1 - r0;
    const int n_steps = (int)((rows + kWK - 1) / kWK);
    const bool is_a = w < 4;
    const bool a_first = o0 < sy.H;                           // this tile's outputs lie in the f1 half (a tile never straddles)
    const bool x_second = sy.X2 != nullptr && i0 >= sy.H;     // ... its inputs in the second half of [X | X2]
    const bool has_t = !EFF && sy.act != GLASS_ACT_NONE;
    // the four row slots of a set: (resource, this thread's byte offset in stage 0, bytes per stage)
    const float* xs = x_second ? sy.X2 : X;
    const int64_t xld = x_second ? sy.ldx2 : ldx;
    const buf_rsrc r_g = make_rsrc(sy.dsrc + r0 * sy.ldd, rows * sy.ldd * 4);
    const buf_rsrc r_t = make_rsrc(has_t ? sy.T + r0 * sy.ldt : sy.dsrc, has_t ? rows * sy.ldt * 4 : 0);  // none: zeros
    const buf_rsrc r_x = make_rsrc(xs + r0 * xld, rows * xld * 4);
    const buf_rsrc r_m = make_rsrc(sy.mask + r0, (!EFF && is_a) ? rows : 0);
    const buf_rsrc rs01 = is_a ? r_g : r_x, rs23 = is_a ? r_t : r_x;
    const int step01 = (int)(kWK * (is_a ? sy.ldd : xld) * 4), step23 = (int)(kWK * (is_a ? sy.ldt : xld) * 4);
    // A wave: row pair g2 = 4 (w & 1) + (l & 3), column quad qa = (l >> 2) + 16 (w >> 1)
    const int g2 = 4 * (w & 1) + (lane & 3), qa = (lane >> 2) + 16 * ((w >> 1) & 1);
    // B wave: row quad gq = 2 (w & 1) + (l & 1), column quad qb = (l >> 1) + 32 ((w - 4) >> 1)
    const int gq = 2 * (w & 1) + (lane & 1), qb = (lane >> 1) + 32 * ((w >> 1) & 1);
    const int a_col = o0 + 4 * qa;
    int off[4];
    if (is_a) {
        const int g_col = a_first ? a_col : a_col - sy.H;
        off[0] = (int)(((2 * g2) * sy.ldd + g_col) * 4);
        off[1] = (int)(((2 * g2 + 1) * sy.ldd + g_col) * 4);
        off[2] = (int)(((2 * g2) * sy.ldt + a_col) * 4);
        off[3] = (int)(((2 * g2 + 1) * sy.ldt + a_col) * 4);
    } else {
        const int xcol = (x_second ? i0 - sy.H : i0) + 4 * qb;
#pragma unroll
        for (int b = 0; b < 4; ++b) off[b] = (int)(((4 * gq + b) * xld + xcol) * 4);
    }
    const int off_m = 2 * g2;
    // A slab with an ODD row count (the last slab of an odd N): its final pair (rows - 1, rows) straddles the end of r_m, and
    // a raw-buffer access that is only partly in range returns 0 for the whole 16 bits — the last real row's label byte is
    // therefore read once up front (wave-uniform) and put back at the one step where this lane holds that pair.
    unsigned last_lbl = 0;
    int strad_step = -1;
    if (!EFF && is_a && (rows & 1)) {
        last_lbl = sy.mask[r1 - 1];
        const int d = (int)(rows - 1) - off_m;
        if (d >= 0 && d % kWK == 0) strad_step = d / kWK;
    }
    const float e_neg = sy.act == GLASS_ACT_RELU ? 0.f : 1.f;  // act'(t) = t > 0 ? 1 : e_neg * exp(t); none: t reads 0, factor 1
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    auto issue = [&](W8Set& S, int step) __attribute__((always_inline)) {
        S.v[0] = buf_load4(rs01, off[0] + step * step01);
        S.v[1] = buf_load4(rs01, off[1] + step * step01);
        S.v[2] = buf_load4(rs23, off[2] + step * step23);
        S.v[3] = buf_load4(rs23, off[3] + step * step23);
        S.mk2 = __builtin_amdgcn_raw_buffer_load_b16(r_m, off_m + step * kWK, 0, 0);
    };
    auto commit = [&](W8Set& S, float* stage, int cstep) __attribute__((always_inline)) {
        S.pin();
        if (is_a) {  // wave-uniform
            float4 gv[2];
            const unsigned mk = (unsigned)S.mk2 | (cstep == strad_step ? last_lbl : 0u);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float4 g = S.v[a];
                if (!EFF) {
                    const float4 t = S.v[2 + a];
                    const float cf = (((mk >> (8 * a)) & 0xffu) != 0) == a_first ? sy.zr : sy.omz;
                    g.x *= cf; g.y *= cf; g.z *= cf; g.w *= cf;
                    g.x *= t.x > 0.f ? 1.f : e_neg * __expf(t.x);
                    g.y *= t.y > 0.f ? 1.f : e_neg * __expf(t.y);
                    g.z *= t.z > 0.f ? 1.f : e_neg * __expf(t.z);
                    g.w *= t.w > 0.f ? 1.f : e_neg * __expf(t.w);
                }
                bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
                gv[a] = g;
            }
            unsigned* A = reinterpret_cast<unsigned*>(stage);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hi, mid, lo;
                split2(f4e(gv[0], e), f4e(gv[1], e), hi, mid, lo);
                const int u = (((g2 >> 2) * kWO + e * 32 + qa) << 2) + (g2 & 3);
                A[u] = hi;
                A[4 * SplitImg<kWO>::kPlane + u] = mid;
                A[8 * SplitImg<kWO>::kPlane + u] = lo;
            }
        } else {
            uint2* B = reinterpret_cast<uint2*>(stage + 4 * SplitImg<kWO>::kUnits);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint2 hi, mid, lo;
                split2(f4e(S.v[0], e), f4e(S.v[1], e), hi.x, mid.x, lo.x);
                split2(f4e(S.v[2], e), f4e(S.v[3], e), hi.y, mid.y, lo.y);
                const int u = (((gq >> 1) * kWI + (qb >> 5) * 128 + e * 32 + (qb & 31)) << 1) + (gq & 1);
                B[u] = hi;
                B[2 * SplitImg<kWI>::kPlane + u] = mid;
                B[4 * SplitImg<kWI>::kPlane + u] = lo;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[rb][cb][k] = 0.f;
    auto stage_mma = [&](const float* cur) __attribute__((always_inline)) {
        const float4* Ai = reinterpret_cast<const float4*>(cur);
        const float4* Bi = Ai + SplitImg<kWO>::kUnits;
        uint4 a[2][3];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) SplitImg<kWO>::frag(Ai, wm * 64 + rb * 32 + j, h, a[rb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            uint4 b[3];
            SplitImg<kWI>::frag(Bi, wn * 64 + cb * 32 + j, h, b);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) split_mma(acc[rb][cb], a[rb], b);
        }
    };

    if (is_a) __builtin_amdgcn_s_setprio(1);  // the A waves (more vector work per cut) first on the matrix pipe: -2 % measured
    W8Set s0, s1, s2;
    auto step_fn = [&](int step, W8Set& nxt_regs) __attribute__((always_inline)) {
        stage_mma(wsm + (step & 1) * kWStageS);
        __builtin_amdgcn_sched_barrier(0);  // (nothing of a later step moves up across these)
        commit(nxt_regs, wsm + ((step + 1) & 1) * kWStageS, step + 1);
        __builtin_amdgcn_sched_barrier(0);
        issue(nxt_regs, step + 4);
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(s0, 0);
    issue(s1, 1);
    issue(s2, 2);
    commit(s0, wsm, 0);
    issue(s0, 3);
    lds_barrier();
    for (int step = 0; step < n_steps; step += 3) {  // whole triples: steps past the slab read zeros and add zeros
        step_fn(step, s1);
        step_fn(step + 1, s2);
        step_fn(step + 2, s0);
    }

    // partial tile, plain [128][256]: slot -> column as the staging laid them out.  Output slot wm*64 + rb*32 + r
    // (r = 8 (k >> 2) + 4 h + (k & 3)) is output 4 r + 2 wm + rb; input slot wn*64 + cb*32 + j is input
    // 4 (32 (wn >> 1) + j) + 2 (wn & 1) + cb: two consecutive inputs per lane
    const int64_t tile_id = ((int64_t)bz * ny + by) * n_slabs + bx;
    float* pw = part_w + tile_id * kWTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int o = 4 * (8 * (k >> 2) + 4 * h + (k & 3)) + 2 * wm + rb;
            *reinterpret_cast<float2*>(pw + o * kWI + 4 * (32 * (wn >> 1) + j) + 2 * (wn & 1)) = make_float2(acc[rb][0][k], acc[rb][1][k]);
        }
    if (by == 0 && part_b) {
        // the 8 A threads of a column quad (l & 3, w & 1) hold partial bias sums of its 4 outputs: combined in a fixed order
        __syncthreads();  // every wave is done with the stages
        float4* red = reinterpret_cast<float4*>(wsm);
        if (is_a) red[qa * 8 + (lane & 3) + 4 * (w & 1)] = bsum;
        __syncthreads();
        if (tid < 32) {
            float4 s = red[tid * 8];
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                const float4 o = red[tid * 8 + r];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *reinterpret_cast<float4*>(part_b + ((int64_t)bz * n_slabs + bx) * kWO + 4 * tid) = s;
        }
    }
}

// dW[o0 + r][i0 + c] (+)= sum over slabs, in slab order; one thread per 4 consecutive inputs.  grid (kWTile / 4 / 256 +
// 1, ny * nz): the last x-block of an input-tile-0 chunk reduces the bias partials.  header[0] != 0: the partials are in
// effective-weight form (tiled_wgrad_kernel<.., EFF>): output tile z of the f1 half = (1-z) S + (2z-1) L, of the f0 half
// = z S - (2z-1) L with S / L the slab sums of chunks zz and nz/2 + zz.
__global__ __launch_bounds__(kWThreads) void tiled_wgrad_reduce_kernel(const float* __restrict__ part_w,
                                                                      const float* __restrict__ part_b,
                                                                      const float* __restrict__ header, int n_slabs,
                                                                      int ny, int nz, float* __restrict__ dW, int64_t lddw,
                                                                      float* __restrict__ db, int accumulate) {
    const int chunk = blockIdx.y, z = chunk / ny, y = chunk % ny;
    const bool eff = header[0] != 0.f;
    const float zr = header[1];
    const int zz = eff ? z % (nz / 2) : z;
    const float cs = !eff ? 1.f : (z < nz / 2 ? 1.f - zr : zr);             // weight of the all-rows sum
    const float cl = !eff ? 0.f : (z < nz / 2 ? 2.f * zr - 1.f : 1.f - 2.f * zr);  // ... of the labeled-rows sum
    if (blockIdx.x == kWTile / 4 / kWThreads) {  // bias block
        if (y != 0 || db == nullptr || threadIdx.x >= kWO) return;
        const float* p = part_b + (int64_t)zz * n_slabs * kWO + threadIdx.x;
        float s = 0.f, l = 0.f;
        for (int b = 0; b < n_slabs; ++b) s += p[(int64_t)b * kWO];
        if (eff) {
            const float* pl = part_b + (int64_t)(nz / 2 + zz) * n_slabs * kWO + threadIdx.x;
            for (int b = 0; b < n_slabs; ++b) l += pl[(int64_t)b * kWO];
        }
        s = cs * s + cl * l;
        float* d = db + z * kWO + threadIdx.x;
        *d = accumulate ? *d + s : s;
        return;
    }
    const int e = (blockIdx.x * kWThreads + threadIdx.x) * 4;  // element of the [128][256] tile
    auto slab_sum = [&](int ch) __attribute__((always_inline)) {
        const float* p = part_w + (int64_t)ch * n_slabs * kWTile + e;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < n_slabs; b += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = b + u < n_slabs ? *reinterpret_cast<const float4*>(p + (int64_t)(b + u) * kWTile) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        return s;
    };
    float4 s = slab_sum(zz * ny + y);
    if (eff) {
        const float4 l = slab_sum((nz / 2 + zz) * ny + y);
        s = make_float4(cs * s.x + cl * l.x, cs * s.y + cl * l.y, cs * s.z + cl * l.z, cs * s.w + cl * l.w);
    }
    float* d = dW + (int64_t)(z * kWO + e / kWI) * lddw + y * kWI + e % kWI;
    if (accumulate) {
        d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w;
    } else {
        d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
    }
}

static inline float* header_of(float* part_b, const TiledWgradGeom& g) { return part_b + g.part_b_floats - kWHeaderFloats; }

void launch_tiled_wgrad_partial(const float* X, int64_t ldx, int64_t N, int64_t O, int64_t I, const WgradSynth& sy,
                                float* part_w, float* part_b, hipStream_t st) {
    const TiledWgradGeom g = wgrad_tiled_geom(N, O, I);
    const dim3 grid(g.n_slabs, g.ny, g.nz);
    float* header = header_of(part_w + g.part_w_floats, g);  // (the bias partials always sit behind the weight partials)
    // comb pair (virtual concatenation as the input, no activation factor, both halves of the output present)
    const bool eff = sy.X2 != nullptr && sy.act == GLASS_ACT_NONE && I == O && g.nz % 2 == 0 && O == 2 * (int64_t)sy.H;
    const bool s3 = tiled_split_products();
    const size_t stage_bytes = (size_t)(s3 ? kWStageS : kWStage) * sizeof(float);
    const size_t lds = 2 * stage_bytes + (eff ? (size_t)ceil_div((int64_t)g.rows_per_slab, (int64_t)kWK) * sizeof(int) : 0);
#define GLASS_TWG(EFFV, S3V)                                                                                          \
    {                                                                                                                \
        if (lds > 64 * 1024)                                                                                         \
            (void)hipFuncSetAttribute((const void*)tiled_wgrad_kernel<true, EFFV, S3V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((tiled_wgrad_kernel<true, EFFV, S3V>), grid, dim3(kWThreads), lds, st, nullptr, 0, X, ldx, N,  \
                           g.rows_per_slab, part_w, part_b, header, sy, 0, g.nz);                                     \
    }
    if (s3) {
        // split form: the all-rows tiles on the eight-wave kernel, the labeled-rows tiles of the effective-weight form on the
        // four-wave kernel (stages without a labeled row are skipped there: short workgroups)
        const int nz8 = eff ? g.nz / 2 : g.nz;
        const dim3 grid8((unsigned)(g.n_slabs * g.ny * nz8));
        const size_t lds8 = 2 * (size_t)kWStageS * sizeof(float);
        if (eff) {
            (void)hipFuncSetAttribute((const void*)tiled_wgrad8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
            hipLaunchKernelGGL((tiled_wgrad8_kernel<true>), grid8, dim3(kW8Threads), lds8, st, X, ldx, N, g.rows_per_slab, part_w,
                               part_b, header, sy, g.n_slabs, g.ny, nz8);
            const dim3 gridl(g.n_slabs, g.ny, g.nz / 2);
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute((const void*)tiled_wgrad_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((tiled_wgrad_kernel<true, true, true>), gridl, dim3(kWThreads), lds, st, nullptr, 0, X, ldx, N,
                               g.rows_per_slab, part_w, part_b, nullptr, sy, g.nz / 2, g.nz);
        } else {
            (void)hipFuncSetAttribute((const void*)tiled_wgrad8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
            hipLaunchKernelGGL((tiled_wgrad8_kernel<false>), grid8, dim3(kW8Threads), lds8, st, X, ldx, N, g.rows_per_slab, part_w,
                               part_b, header, sy, g.n_slabs, g.ny, nz8);
        }
    } else if (eff) {
        GLASS_TWG(true, false)
    } else {
        GLASS_TWG(false, false)
    }
#undef GLASS_TWG
}

void launch_tiled_wgrad_reduce(const float* part_w, const float* part_b, int64_t N, int64_t O, int64_t I, float* dW,
                               int64_t lddw, float* db, int accumulate, hipStream_t st) {
    const TiledWgradGeom g = wgrad_tiled_geom(N, O, I);
    hipLaunchKernelGGL(tiled_wgrad_reduce_kernel, dim3(kWTile / 4 / kWThreads + 1, g.ny * g.nz), dim3(kWThreads), 0, st,
                       part_w, part_b, header_of(const_cast<float*>(part_b), g), g.n_slabs, g.ny, g.nz, dW, lddw, db, accumulate);
}

}  // namespace glass
