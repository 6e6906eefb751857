// 3x3 stride-1 pad-1 forward of the dense layers (K = 128 -> N = 32), third generation: producer / consumer waves.
//
// The ring kernel (conv3x3_ring.hip) runs its phases in series -- stage (BN + ReLU of the new rows into the ring), barrier, 72 MFMAs
// per sub-tile, store epilogue, barrier -- with all eight waves in lock step: the matrix pipe's share of a step is a third and no
// load is in flight while the rows are staged (2.0-2.2x the stream time of its own bytes, profiles/r04_*).  Its LDS is full (ring +
// nine weight slices), so the next rows could not be staged beside the multiplication.
//
// Here the two waves of every SIMD have different jobs (roles split by wave number >= 4: MI355X_MICROARCH.md, "Two waves per SIMD"):
//   * waves 4-7, PRODUCERS: keep DEPTH steps of new input rows in flight as plain 16-byte loads (registers), apply BN + ReLU and
//     write the rows into the ring; their vector work issues beside the partner wave's MFMAs;
//   * waves 0-3, CONSUMERS: one 32-pixel sub-tile per wave and step = 72 back-to-back MFMAs fed by two ds_read_b128 per MFMA (the
//     LDS array's rate: 256 B/clk), then the store epilogue (v_permlane32_swap -> 8 consecutive channels per lane, 16-byte stores,
//     per-lane channel sums).
//   * ONE barrier per step: between barrier j and j+1 the consumers multiply step j while the producers write the rows of step j+1
//     into ring slots step j does not read (ring of 2R+2 rows), and the loads of steps j+2.. stay in flight across the barrier.
// What makes the room: 256-byte ring pixels / weight rows with the 16-byte chunks XOR-swizzled by the pixel (row) index instead of
// the 272-byte padded pitch (conflict-free for the lane groups of ds_read_b128 and ds_write_b128 alike), and short steps
// (R x P <= 128 flat pixels = at most four sub-tiles): 73.7 KB of weights + an 8-row ring of 42-pixel rows = 160.3 KB.
//
// Row space.  The column tiles (image x tile) are stacked into ONE virtual row space with a zero separator row between them
// (stride Hs = H + 1).  A workgroup owns the virtual rows [v0, v1); its step j reads virtual rows [a + jR - 2, a + jR + R),
// writes the output rows [a + jR - 1, a + jR + R - 1) that lie inside its range, and the producers load exactly the R rows
// [a + jR, a + jR + R) for it -- every step alike, image boundaries included (the separator row is the zero padding of both
// neighbours).  A pass starts with a warm-up step (j = -1, rows [a - R, a): the two rows above the range when it starts inside an
// image; dummy loads otherwise).  The range is walked in two passes from a per-workgroup offset, as in the ring kernel (256
// workgroups starting at row 0 of their own images read addresses that differ by multiples of the image size at the same moment).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int NT = 512;                  // threads: waves 0-3 consume, waves 4-7 produce
constexpr int NPT = 256;                 // producer threads
constexpr int PXB = 256;                 // bytes per ring pixel / weight row: 128 bf16, 16 chunks of 16 B, chunk c of row r at slot c ^ (r & 15)
constexpr int W_ROWS = 9 * 32;
constexpr int W_BYTES = W_ROWS * PXB;    // 73728
constexpr int RING_PX_MAX = (160 * 1024 - W_BYTES) / PXB;     // 352 (the statistics scratch re-uses the weight area at the end)

struct PcGeo {
  int H, W, Wt, P, R, NR, Q;    // Wt: column-tile width, P = Wt + 2, NR = 2R + 2 ring rows, Q = NR * P ring pixels (+2 spare)
  int Hs;                       // H + 1: virtual rows per column tile
  int ntx, ntx_shift;           // column tiles per image (1 or 2)
  unsigned V;                   // virtual rows in all = B * ntx * Hs
  unsigned mP, mHs;             // ceil(2^32 / P), ceil(2^32 / Hs)
  int nwg;
  int rot;                      // 1: two passes from a per-workgroup offset
};

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_sum(float v) {      // sum over the 32 lanes of each wave half
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v + __shfl_xor(v, 16);
}

#ifdef CX_PC_STAMPS
// diagnostic build (scratch/stamps_pc.py): s_memtime sums per phase, wave 0 (consumer) and wave 4 (producer) of each workgroup
__device__ unsigned long long pc_stamps[1024 * 16];
__device__ __forceinline__ unsigned long long pstamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PSTAMP(i) { const unsigned long long t_ = pstamp(); st_acc[i] += t_ - st_prev; st_prev = t_; }
#else
#define PSTAMP(i)
#endif

// The step barrier, raw: __syncthreads() would add a full s_waitcnt vmcnt(0) -- the consumers' output stores and the producers' loads
// in flight have nothing to do with the hand-off, which only needs the producers' LDS writes retired (lgkmcnt) before they arrive.
__device__ __forceinline__ void bar_after_lds_writes() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void bar_plain() { asm volatile("s_barrier" ::: "memory"); }

template <int NCH, int DEPTH>
__global__ __launch_bounds__(NT, 1) void conv3x3_pc_fwd_kernel(const bf16* __restrict__ x, int ldx, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, const bf16* __restrict__ wpk,
                                                              bf16* __restrict__ y, int ldy, float* stat_sum, float* stat_sq,
                                                              int stat_replicas, int stat_rstride, int stat_det, const PcGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                             // [9*32][256 B], swizzled
  char* ring = smem + W_BYTES;                                 // [Q + 2][256 B], swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int P = g.P, R = g.R, NR = g.NR, Q = g.Q, W = g.W, H = g.H, Wt = g.Wt, Hs = g.Hs;

  // ---- one-time setup: weights [tap][n][k] -> LDS rows of 256 B (chunk c of row r at slot c ^ (r & 15)), ring zeroed
  for (int i = tid; i < W_ROWS * 16; i += NT) {
    const int row = i >> 4, c = i & 15;
    *reinterpret_cast<uint4*>(wl + row * PXB + ((c ^ (row & 15)) << 4)) = *reinterpret_cast<const uint4*>(wpk + (size_t)row * 128 + c * 8);
  }
  for (int i = tid; i < (Q + 2) * (PXB / 16); i += NT) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const unsigned v0 = (unsigned)(((unsigned long long)blockIdx.x * g.V) / (unsigned)g.nwg);
  const unsigned v1 = (unsigned)(((unsigned long long)(blockIdx.x + 1) * g.V) / (unsigned)g.nwg);
  const unsigned nrows = v1 - v0;
  const unsigned rot = (g.rot && nrows > (unsigned)(2 * R)) ? (blockIdx.x * 37u) % nrows : 0u;

  float s1[2][8], s2[2][8];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[cc][j] = s2[cc][j] = 0.f;

  // rows [a, e) of a pass; a step grid of its own: step j loads rows [a + jR, a + jR + R); steps 0 .. J-1 cover the output rows
  // [a - 1, a + JR - 1) >= [a, e).  Ring slot of virtual row a + r (r >= -R - 2): (r + 2 NR) mod NR, tracked incrementally by both
  // roles.  The roles are the OUTER branch (each walks both passes itself): merged inside the pass loop, the compiler's wait-count
  // bookkeeping carried the producers' loads in flight into the consumers' code and made every sub-tile wait for the wave's own
  // output stores (s_waitcnt vmcnt(0)).
  auto pass_range = [&](int pass, unsigned& a, unsigned& e, int& J) __attribute__((always_inline)) {
    a = pass == 0 ? v0 + rot : v0;
    e = pass == 0 ? v1 : v0 + rot;
    J = e > a ? (int)((e - a + 1 + R - 1) / R) : 0;
  };
  if (wave >= 4) {
    // ================================================================================================= producers
    const int ptid = tid - NPT;
    const int c8 = ptid & 15;                                // this thread's 16-byte channel chunk, the same for every slot
    float csc[8], csh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { csc[j] = sc[c8 * 8 + j]; csh[j] = sh[c8 * 8 + j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(csc[j]), "v"(csh[j]));      // retired before the loop (see the ring kernel)
    // chunk slot i of this thread: chunk id ptid + 256 i -> (row of the step's R new rows, ring position in the row); the rest of a
    // chunk's address is per ROW: lane r of every producer wave works out row r of the step once (tile, image row, validity, byte
    // offset of the row's ring position 0) and the chunks fetch their row's word with ds_bpermute -- per chunk: one cross-lane read,
    // one add, three bit tests (the divisions / 32-bit multiplies per chunk were a third of the producers' instructions)
    const int cpr = P * 16;
    int crow4[NCH], posoff[NCH];
    uint32_t coff[NCH];
    unsigned m0 = 0, m1 = 0, mrow = 0;       // bit i: chunk i's column exists in a tile with tx = 0 / tx = 1; chunk i is inside the R rows
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cid = ptid + NPT * i;
      const int cr = cid / cpr, cp = (cid - cr * cpr) >> 4;
      crow4[i] = cr * 4;
      posoff[i] = cr * P + cp;
      coff[i] = ((uint32_t)cp * (uint32_t)ldx + (uint32_t)c8 * 8u) * 2u;
      mrow |= cr < R ? (1u << i) : 0u;
      m0 |= ((unsigned)(cp - 1) < (unsigned)W) ? (1u << i) : 0u;
      m1 |= ((unsigned)(Wt - 1 + cp) < (unsigned)W) ? (1u << i) : 0u;
    }
    m0 &= mrow;
    m1 &= mrow;
    const char* __restrict__ xb = reinterpret_cast<const char*>(x);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      unsigned a, e;
      int J;
      pass_range(pass, a, e, J);
      if (J == 0) continue;
      uint4 pre[DEPTH][NCH];
      unsigned pvm[DEPTH];
#ifdef CX_PC_STAMPS
      unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = pstamp();
#endif
      // rows [a + jR, a + jR + R) -> registers; chunks outside the images (separator rows, halo columns past the image edge, rows
      // past the tensor or above the two rows a warm-up step needs) read offset 0 and are zeroed when staged
      auto issue = [&](uint4 (&pr)[NCH], unsigned& pm, int j) __attribute__((always_inline)) {
        const unsigned v = a + (unsigned)(j * R) + (unsigned)lane;      // (wraps for j = -1 and a < R: such rows fail the v < V test)
        const unsigned t = __umulhi(v, g.mHs), yy = v - t * (unsigned)Hs;
        const unsigned b = t >> g.ntx_shift, tx = t & (unsigned)(g.ntx - 1);
        const unsigned rok = (unsigned)(j < J) & (unsigned)(lane < R) & (unsigned)(v < g.V) & (unsigned)(yy < (unsigned)H) &
                             ((unsigned)(j >= 0) | (unsigned)(v + 2u >= a));
        // byte offset of ring position 0 of this row (image column tx*Wt - 1: wraps for tx = 0, undone by a valid chunk's coff)
        const uint32_t roff = (((b * (uint32_t)H + yy) * (uint32_t)W + tx * (uint32_t)Wt - 1u) * (uint32_t)ldx) * 2u;
        const int word = (int)(roff | rok | (tx << 1));        // (roff is a multiple of 16: ldx % 8 == 0)
        pm = 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute(crow4[i], word);
          const unsigned ok = w & (((w & 2u) ? m1 : m0) >> i) & 1u;
          pm |= ok << i;
          pr[i] = *reinterpret_cast<const uint4*>(xb + (size_t)(ok ? (w & ~15u) + coff[i] : 0u));
        }
      };
      auto stage = [&](uint4 (&pr)[NCH], unsigned pm, int sb) __attribute__((always_inline)) {
        const int sbp = sb * P, wrap_from = (NR - sb) * 4;     // chunk rows >= NR - sb wrap to the ring's start
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          // straight-line code (a per-chunk `if` becomes an exec-mask branch, and behind each the compiler re-waits for loads
          // -- of the set just issued, too): a chunk past the step's rows goes to the spare pixel Q + 1, which only invalid
          // outputs read
          uint4 o = cx_affine_relu8(pr[i], csc, csh);
          const unsigned keep = 0u - ((pm >> i) & 1u);
          o.x &= keep; o.y &= keep; o.z &= keep; o.w &= keep;
          int pos = sbp + posoff[i] - (crow4[i] >= wrap_from ? Q : 0);
          pos = ((mrow >> i) & 1u) ? pos : Q + 1;
          *reinterpret_cast<uint4*>(ring + pos * PXB + ((c8 ^ (pos & 15)) << 4)) = o;
        }
      };
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {                        // (set by set: the loop's counted waits assume the sets were requested in order)
        issue(pre[d], pvm[d], -1 + d);
        __builtin_amdgcn_sched_barrier(0);
      }
      int sb = (2 * NR - R) % NR;                              // slot of row a - R (the warm-up step's first row)
      auto pstep = [&](int j, auto KI) __attribute__((always_inline)) {
        constexpr int k = decltype(KI)::value;
        PSTAMP(0)
#ifdef CX_PC_STAMPS
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH * (DEPTH - 1)) : "memory");
        PSTAMP(1)                                              // wait for the oldest set
#endif
        stage(pre[k], pvm[k], sb);
        PSTAMP(2)
        issue(pre[k], pvm[k], j + DEPTH);                      // in flight across the next DEPTH barriers
        sb += R;
        if (sb >= NR) sb -= NR;
        PSTAMP(3)
        __syncthreads();
        PSTAMP(4)                                // barrier j: the rows of step j are in the ring
      };
      for (int j = -1; j < J; j += DEPTH) {
        pstep(j, std::integral_constant<int, 0>());
        if (DEPTH > 1) { if (j + 1 >= J) break; pstep(j + 1, std::integral_constant<int, 1 % DEPTH>()); }
        if (DEPTH > 2) { if (j + 2 >= J) break; pstep(j + 2, std::integral_constant<int, 2 % DEPTH>()); }
      }
      bar_plain();                                             // the pass is over: the ring is rebuilt by the next one
#ifdef CX_PC_STAMPS
      if (tid == 256 && blockIdx.x < 1024) {
        for (int i = 0; i < 5; ++i) pc_stamps[blockIdx.x * 16 + 8 + i] += st_acc[i];
        pc_stamps[blockIdx.x * 16 + 15] += (unsigned long long)(J + 1);
      }
#endif
    }
  } else {
    // ================================================================================================= consumers
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      unsigned a, e;
      int J;
      pass_range(pass, a, e, J);
      if (J == 0) continue;
      const int nsub = (R * P + 31) / 32;
      // weight fragment of tap t, k-step ks: row t*32 + lrow, chunk 2 ks + lh -> slot (2 ks + lh) ^ (lrow & 15)
      const char* wrow = wl + lrow * PXB;
      const int wsw = ((lrow & 15) ^ lh) << 4;                 // ^ (ks << 5) per k-step
      int sb = (2 * NR - R - 2) % NR;                          // slot of row a + jR - 2 for j = -1
#ifdef CX_PC_STAMPS
      unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = pstamp();
#endif
      bar_plain();                                             // barrier -1 (warm-up rows)
      for (int j = 0; j < J; ++j) {
        sb += R;
        if (sb >= NR) sb -= NR;
        PSTAMP(0)
        bar_plain();                                           // barrier j (this wave's fragment reads of step j-1 were consumed by its MFMAs)
        PSTAMP(1)
        const int ws = sb * P;
        for (int s = wave; s < nsub; s += 4) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const int m = s * 32 + lrow;
          const int pix = min(m, R * P - 1);
          // ring pixel of tap (dy, dx): ws + pix + dy*P + dx (mod Q; + dx may run into the two spare pixels: invalid outputs only)
          int pb[3];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            int f = ws + pix + dy * P;
            if (f >= Q) f -= Q;
            if (f >= Q) f -= Q;
            pb[dy] = f;
          }
          constexpr int G = 3, NG = 72 / G;
          bf16x8 fa[3][G], fb[3][G];
          auto load_grp = [&](int gi, bf16x8 (&A)[G], bf16x8 (&B)[G]) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < G; ++jj) {
              const int f = gi * G + jj, t = f >> 3, ks = f & 7;
              const int dy = t / 3, dx = t - dy * 3;
              A[jj] = *reinterpret_cast<const bf16x8*>(wrow + t * 32 * PXB + (wsw ^ (ks << 5)));
              const int p = pb[dy] + dx;
              B[jj] = *reinterpret_cast<const bf16x8*>(ring + p * PXB + ((((p & 15) ^ lh) << 4) ^ (ks << 5)));
            }
          };
          load_grp(0, fa[0], fb[0]);
          load_grp(1, fa[1], fb[1]);
#pragma unroll
          for (int gi = 0; gi < NG; ++gi) {
            if (gi + 2 < NG) load_grp(gi + 2, fa[(gi + 2) % 3], fb[(gi + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < G; ++jj)
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gi % 3][jj], fb[gi % 3][jj], acc, 0, 0, 0);   // D[row = out channel][col = pixel]
            __builtin_amdgcn_sched_barrier(0);
          }
#ifdef CX_PC_STAMPS
          asm volatile("" ::"v"(acc[0]));
#endif
          PSTAMP(2)
          const unsigned oy = __umulhi((unsigned)m, g.mP), ox = (unsigned)m - oy * (unsigned)P;
          const unsigned v = a + (unsigned)(j * R) - 1u + oy;                  // (a = 0, j = 0, oy = 0 wraps: fails v >= a)
          const unsigned t = __umulhi(v, g.mHs), yy = v - t * (unsigned)Hs;
          const int b = (int)(t >> g.ntx_shift), xc = (int)(t & (unsigned)(g.ntx - 1)) * Wt + (int)ox;
          const bool valid = m < R * P && ox < (unsigned)Wt && xc < W && v >= a && v < e && yy < (unsigned)H;
          bf16* yrow = y + (valid ? ((size_t)(b * H + (int)yy) * W + xc) * ldy : (size_t)0);
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            // registers 8cc..8cc+3 / 8cc+4..8cc+7: channels 16cc + 4*lh + e / 16cc + 8 + 4*lh + e; the swap of the upper half of
            // the first group with the lower half of the second leaves channels 8*(2cc+lh) .. +7 of this lane's pixel
            U128 o;
            float tv[8];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * cc + r4]), __float_as_uint(acc[8 * cc + 4 + r4]),
                                                               false, false);
              tv[r4] = __uint_as_float(sw[0]);
              tv[4 + r4] = __uint_as_float(sw[1]);
            }
            o.u = cx_pack8_stats(tv, valid, true, s1[cc], s2[cc]);
            if (valid) *reinterpret_cast<uint4*>(yrow + 8 * (2 * cc + lh)) = o.u;
          }
          PSTAMP(3)
        }
      }
      bar_plain();                                             // the pass is over
#ifdef CX_PC_STAMPS
      if (tid == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 4; ++i) pc_stamps[blockIdx.x * 16 + i] += st_acc[i];
        pc_stamps[blockIdx.x * 16 + 7] += (unsigned long long)J;
      }
#endif
    }
  }

  if (stat_sum) {
    float* scratch = reinterpret_cast<float*>(wl);               // the weight slices are no longer read
    wg_stat_begin<NT / 64>(scratch, 32, tid, NT);
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a_ = half_sum(s1[cc][j]);
        const float c_ = half_sum(s2[cc][j]);
        if (lrow == 8 * cc + j) { t1 = a_; t2 = c_; }
      }
    if (lrow < 16) {
      const int n = 8 * (2 * (lrow >> 3) + lh) + (lrow & 7);
      wg_stat_put(scratch, 32, wave, n, t1, t2);
    }
    wg_stat_end<NT / 64>(scratch, 32, tid, NT, stat_sum, stat_sq, stat_det, (int)blockIdx.x, stat_replicas, stat_rstride, 0, 32);
  }
}

template <int NCH, int DEPTH>
int launch_pc(const CxConv& p, hipStream_t st, const PcGeo& g) {
  const size_t smem = W_BYTES + (size_t)(g.Q + 2) * PXB;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pc_fwd_kernel<NCH, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  if (const int e = stat_rows_check(p, g.nwg)) return e;
  CX_KTAG("conv3x3_pc_fwd_kernel<%d, %d>", NCH, DEPTH);
  hipLaunchKernelGGL((conv3x3_pc_fwd_kernel<NCH, DEPTH>), dim3(g.nwg), dim3(NT), smem, st, (const bf16*)p.x, p.ldx, p.pa, p.pb,
                     (const bf16*)p.w, (bf16*)p.y, p.ldy, p.stat_sum, p.stat_sq, p.stat_replicas, p.stat_rstride, p.stat_det, g);
  return launch_status();
}

}  // namespace

#ifdef CX_PC_STAMPS
extern "C" int dbg_pc_stamps(unsigned long long* host, int n_words) {      // host == nullptr: zero the sums
  if (!host) {
    static unsigned long long z[1024 * 16];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(pc_stamps), z, sizeof(z), 0, hipMemcpyHostToDevice);
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pc_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// Eligibility + launch, called from cx_conv_gemm ahead of the ring kernel.  CxConv.kernel_hint form 7 pins the ring kernel (tests, A/B).
int cx_try_pc_fwd(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (((p.kernel_hint >> 8) & 0xff) == 8) return 0;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1 || p.tstride > 1) return 0;
  if (p.K != 128 || p.N != 32 || p.prologue != CX_PRO_AFFINE_RELU || p.epilogue != CX_EPI_STORE || p.accumulate) return 0;
  if (p.W < 4 || p.H < 1) return 0;
  if ((unsigned long long)p.B * p.H * p.W * (unsigned long long)p.ldx * 2ull >= (1ull << 32)) return 0;      // 32-bit chunk offsets in the kernel
  PcGeo g;
  g.H = p.H; g.W = p.W;
  g.ntx = (p.W >= 64 && p.W % 2 == 0) ? 2 : 1;
  g.ntx_shift = g.ntx - 1;
  g.Wt = p.W / g.ntx;
  g.P = g.Wt + 2;
  g.Hs = p.H + 1;
  // largest R with R*P <= 128 flat pixels (four sub-tiles, eight chunks per producer thread) and the 2R+2-row ring in LDS
  int R = 128 / g.P;
  while (R >= 1 && (2 * R + 2) * g.P + 2 > RING_PX_MAX) --R;
  if (R < 1) return 0;
  if (R > g.Hs) R = g.Hs;
  g.R = R; g.NR = 2 * R + 2; g.Q = g.NR * g.P;
  const unsigned long long V = (unsigned long long)p.B * g.ntx * g.Hs;
  if (V * (unsigned long long)g.Hs >= (1ull << 32) || V + 4096 >= (1ull << 31)) return 0;       // exact multiply-high divisions, int step arithmetic
  g.V = (unsigned)V;
  g.mP = 0xffffffffu / (unsigned)g.P + 1u;
  g.mHs = 0xffffffffu / (unsigned)g.Hs + 1u;
  // one workgroup per CU; fewer when the rows would not fill two steps each
  int nwg = 256;
  if (V < (unsigned long long)nwg * 2 * R) nwg = (int)((V + 2 * R - 1) / (2 * R));
  if (nwg < 1) nwg = 1;
  g.nwg = nwg;
  g.rot = 1;
  *handled = true;
  const int need = (R * g.P * 16 + NPT - 1) / NPT;
  if (need <= 4) return launch_pc<4, 3>(p, st, g);
  if (need <= 6) return launch_pc<6, 2>(p, st, g);
  return launch_pc<8, 2>(p, st, g);
}
