// 3x3 stride-1 pad-1 forward of the dense layers (K = 128 -> N = 32), third generation: producer / consumer waves, the weights in
// registers, the input channels split over the consumer waves.
//
// The ring kernel (conv3x3_ring.hip) runs its phases in series -- stage (BN + ReLU of the new rows into the ring), barrier, 72 MFMAs
// per sub-tile, store epilogue, barrier -- with all eight waves in lock step (2.0-2.2x the stream time of its own bytes,
// profiles/r04_*).  A first producer / consumer form of it (four waves staging, four multiplying one sub-tile each, the weights
// still in LDS; git history of this file) hid every load and ran no faster: with N = 32 an MFMA needs 1 KB of weights AND 1 KB of
// pixels from LDS, four waves at two ds_read_b128 per MFMA saturate the LDS array (256 B/clk) on their own, the producers' writes
// queue behind them (stamps: 52 cycles per MFMA, 430 per staged chunk; profiles/r05_stamps_pc.txt).
//
// So the contraction is split the other way:
//   * waves 0-3, CONSUMERS: wave w owns the input channels [32w, 32w + 32) of ALL nine taps -- 18 weight fragments = 72 VGPRs,
//     loaded once per workgroup -- and multiplies them with every pixel of the step (four 32-pixel sub-tiles, four accumulators):
//     ONE ds_read_b128 per MFMA, no weights in LDS at all.  The four partial sums of a sub-tile meet through LDS (64 KB of the room
//     the weights left): wave t adds the partials of sub-tile t in wave order (a fixed order: results are reproducible) and runs
//     the store epilogue (v_permlane32_swap -> 8 consecutive channels per lane, 16-byte stores, per-lane channel sums);
//   * waves 4-7, PRODUCERS: keep DEPTH steps of new input rows in flight as plain 16-byte loads (registers), apply BN + ReLU and
//     write the rows into the ring (272-byte pixels: conflict-free without address arithmetic); their vector work issues beside
//     the partner wave's MFMAs (roles split by wave number >= 4: MI355X_MICROARCH.md, "Two waves per SIMD");
//   * two barriers per step: A_j (the rows of step j are staged) and B_j (the partial sums of step j are written).  Between A_j
//     and B_j the consumers multiply step j, between B_j and A_j+1 they add and store it; the producers stage the rows of step
//     j + 1 meanwhile (half before B_j, half after) into ring slots step j does not read (ring of 2R + 2 rows), and the loads
//     of steps j + 2.. stay in flight across the barriers (raw s_barrier: no vmcnt drain).
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
constexpr int XP = 272;                  // bytes per ring pixel: 128 bf16 + 16 pad (conflict-free ds_read_b128 / ds_write_b128)
constexpr int PART_BYTES = 4 * 4 * 4 * 1024;   // partial sums: [wave 4][sub-tile 4][register quad 4][lane 64][16 B] = 64 KB
constexpr int RING_PX_MAX = (160 * 1024 - PART_BYTES) / XP;     // 361

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

#ifndef CX_PC_ABL
#define CX_PC_ABL 0      // timing ablations of diagnostic builds (results wrong): 1 no MFMAs, 2 no output stores, 4 no input loads, 8 no BN + ReLU
#endif
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

// The step barriers, raw: __syncthreads() adds a full s_waitcnt vmcnt(0) where stores are pending -- the consumers' output stores
// have nothing to do with the hand-offs, which only need this wave's LDS writes retired (lgkmcnt) before it arrives.
__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NCH, int DEPTH>
__global__ __launch_bounds__(NT, 1) void conv3x3_pc_fwd_kernel(const bf16* __restrict__ x, int ldx, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, const bf16* __restrict__ wpk,
                                                              bf16* __restrict__ y, int ldy, float* stat_sum, float* stat_sq,
                                                              int stat_replicas, int stat_rstride, int stat_det, const PcGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* part = smem;                                           // partial sums, 64 KB (the statistics scratch at the end)
  char* ring = smem + PART_BYTES;                              // [Q + 2][272 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int P = g.P, R = g.R, NR = g.NR, Q = g.Q, W = g.W, H = g.H, Wt = g.Wt, Hs = g.Hs;

  // ---- one-time setup: ring zeroed (the pad columns at the image edges are never written again)
  for (int i = tid; i < (Q + 2) * (XP / 16); i += NT) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
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
  // bookkeeping carries the producers' loads in flight into the consumers' code (every sub-tile then waits for the wave's own
  // output stores: s_waitcnt vmcnt(0)).
  auto pass_range = [&](int pass, unsigned& a, unsigned& e, int& J) __attribute__((always_inline)) {
    a = pass == 0 ? v0 + rot : v0;
    e = pass == 0 ? v1 : v0 + rot;
    J = e > a ? (int)((e - a + 1 + R - 1) / R) : 0;
  };
  // Barriers of a pass, the same 2J + 3 for both roles: A_-1, then B_j-1 and A_j for j = 0 .. J-1, B_J-1, end of pass.
  if (wave >= 4) {
    // ================================================================================================= producers
#ifndef CX_PC_PRIO
#define CX_PC_PRIO 2
#endif
    // static priority for the staging waves: beside a partner that streams MFMAs (each holds the SIMD's vector issue for 8 of its 32
    // cycles, and the older wave wins every arbitration) their ~45 vector instructions per chunk crawled at 19 cycles each
    __builtin_amdgcn_s_setprio(CX_PC_PRIO);
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
    // one add, three bit tests (divisions / 32-bit multiplies per chunk were a third of the producers' instructions)
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
    char* ringc = ring + c8 * 16;
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
          if (CX_PC_ABL & 4) pr[i] = make_uint4(w, w, w, w);
          else pr[i] = *reinterpret_cast<const uint4*>(xb + (size_t)(ok ? (w & ~15u) + coff[i] : 0u));
        }
      };
      // chunks [I0, I1) of a set -> ring.  Straight-line code (a per-chunk `if` becomes an exec-mask branch, and behind each the
      // compiler re-waits for loads -- of the set just issued, too): a chunk past the step's rows goes to the spare pixel Q + 1,
      // which only invalid outputs read
      auto stage = [&](uint4 (&pr)[NCH], unsigned pm, int sb, auto I0, auto I1) __attribute__((always_inline)) {
        const int sbp = sb * P, wrap_from = (NR - sb) * 4;     // chunk rows >= NR - sb wrap to the ring's start
#pragma unroll
        for (int i = decltype(I0)::value; i < decltype(I1)::value; ++i) {
          uint4 o = (CX_PC_ABL & 8) ? pr[i] : cx_affine_relu8(pr[i], csc, csh);
          const unsigned keep = 0u - ((pm >> i) & 1u);
          o.x &= keep; o.y &= keep; o.z &= keep; o.w &= keep;
          int pos = sbp + posoff[i] - (crow4[i] >= wrap_from ? Q : 0);
          pos = ((mrow >> i) & 1u) ? pos : Q + 1;
          *reinterpret_cast<uint4*>(ringc + pos * XP) = o;
        }
      };
      using I_0 = std::integral_constant<int, 0>;
      using I_H = std::integral_constant<int, NCH / 2>;
      using I_N = std::integral_constant<int, NCH>;
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {                        // (set by set: the loop's counted waits assume the sets were requested in order)
        issue(pre[d], pvm[d], -1 + d);
        __builtin_amdgcn_sched_barrier(0);
      }
      int sb = (2 * NR - R) % NR;                              // slot of row a - R (the warm-up step's first row)
      // warm-up rows, then barrier A_-1
      stage(pre[0], pvm[0], sb, I_0{}, I_N{});
      issue(pre[0], pvm[0], -1 + DEPTH);
      sb += R;
      if (sb >= NR) sb -= NR;
      bar_lds();
      auto pstep = [&](int j, auto KI) __attribute__((always_inline)) {
        constexpr int k = decltype(KI)::value;
        PSTAMP(0)
        stage(pre[k], pvm[k], sb, I_0{}, I_H{});               // beside the consumers' multiplication of step j - 1
        PSTAMP(1)
        bar_lds();                                             // B_j-1
        PSTAMP(2)
        stage(pre[k], pvm[k], sb, I_H{}, I_N{});               // beside their sums and stores
        PSTAMP(3)
        issue(pre[k], pvm[k], j + DEPTH);                      // in flight across the next 2 DEPTH barriers
        sb += R;
        if (sb >= NR) sb -= NR;
        PSTAMP(4)
        bar_lds();                                             // A_j: the rows of step j are in the ring
        PSTAMP(5)
      };
      for (int j = 0; j < J; j += DEPTH) {
        pstep(j, std::integral_constant<int, 1 % DEPTH>());
        if (DEPTH > 1) { if (j + 1 >= J) break; pstep(j + 1, std::integral_constant<int, 2 % DEPTH>()); }
        if (DEPTH > 2) { if (j + 2 >= J) break; pstep(j + 2, std::integral_constant<int, 3 % DEPTH>()); }
      }
      bar_lds();                                               // B_J-1
      bar_lds();                                               // the pass is over: the ring is rebuilt by the next one
#ifdef CX_PC_STAMPS
      if (tid == 256 && blockIdx.x < 1024) {
        for (int i = 0; i < 6; ++i) pc_stamps[blockIdx.x * 16 + 8 + i] += st_acc[i];
        pc_stamps[blockIdx.x * 16 + 15] += (unsigned long long)J;
      }
#endif
    }
  } else {
    // ================================================================================================= consumers
    // weight fragments of this wave's 32 input channels: tap t, k-step ks (16 channels): A[row = out channel lrow][k = 8 lh .. + 8]
    bf16x8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[t][ks] = *reinterpret_cast<const bf16x8*>(wpk + (size_t)(t * 32 + lrow) * 128 + wave * 32 + ks * 16 + lh * 8);
    const int choff = (wave * 4 + lh) * 16;                    // this lane's 16 bytes of a ring pixel for k-step 0 (+ 32 for k-step 1)
    // partial sums: register quad q of sub-tile tt from wave w at part[((w*4 + tt)*4 + q)*1024 + lane*16]
    char* pw = part + wave * 16 * 1024 + lane * 16;            // this wave's partials (writer)
    const char* pr_ = part + wave * 4 * 1024 + lane * 16;      // sub-tile `wave` of every wave (reader): + w * 16 KB + q * 1 KB
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      unsigned a, e;
      int J;
      pass_range(pass, a, e, J);
      if (J == 0) continue;
      int sb = (2 * NR - R - 2) % NR;                          // slot of row a + jR - 2 for j = -1
#ifdef CX_PC_STAMPS
      unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = pstamp();
#endif
      bar_lds();                                               // A_-1 (warm-up rows)
      bar_lds();                                               // B_-1
      for (int j = 0; j < J; ++j) {
        sb += R;
        if (sb >= NR) sb -= NR;
        PSTAMP(0)
        bar_lds();                                             // A_j
        PSTAMP(1)
        const int ws = sb * P;
        f32x16 acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
        // ring pixel of tap (dy, dx) of sub-tile tt: ws + pix + dy*P + dx (mod Q; + dx may run into the two spare pixels: invalid
        // outputs only); byte address of this lane's k-step-0 chunk
        const char* bp[4][3];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int pix = min(tt * 32 + lrow, R * P - 1);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            int f = ws + pix + dy * P;
            if (f >= Q) f -= Q;
            if (f >= Q) f -= Q;
            bp[tt][dy] = ring + f * XP + choff;
          }
        }
        // 18 (tap, k-step) groups of four MFMAs (one per sub-tile, the same weight fragment); the reads of group g + 1 go out before
        // the MFMAs of group g (128 cycles ahead; a third register set does not fit beside 72 weight and 64 accumulator registers)
        constexpr int NG = 18, NS = 2;
        bf16x8 fb[NS][4];
        auto load_grp = [&](int gi, bf16x8 (&B)[4]) __attribute__((always_inline)) {
          const int t = gi >> 1, ks = gi & 1, dy = t / 3, dx = t - dy * 3;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) B[tt] = *reinterpret_cast<const bf16x8*>(bp[tt][dy] + dx * XP + ks * 32);
        };
#pragma unroll
        for (int d = 0; d < NS - 1; ++d) load_grp(d, fb[d]);
#pragma unroll
        for (int gi = 0; gi < ((CX_PC_ABL & 1) ? 0 : NG); ++gi) {
          if (gi + NS - 1 < NG) load_grp(gi + NS - 1, fb[(gi + NS - 1) % NS]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[gi >> 1][gi & 1], fb[gi % NS][tt], acc[tt], 0, 0, 0);   // D[row = out channel][col = pixel]
          __builtin_amdgcn_sched_barrier(0);
        }
        PSTAMP(2)
        // partial sums -> LDS (lane-linear 16-byte pieces: conflict-free)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(pw + (tt * 4 + q) * 1024) = f32x4{acc[tt][4 * q], acc[tt][4 * q + 1], acc[tt][4 * q + 2], acc[tt][4 * q + 3]};
        PSTAMP(3)
        bar_lds();                                             // B_j
        PSTAMP(4)
        // sub-tile `wave`: the four partials in wave order
        f32x16 sum;
        {
          f32x4 pq[4][4];
#pragma unroll
          for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) pq[w][q] = *reinterpret_cast<const f32x4*>(pr_ + w * 16 * 1024 + q * 1024);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = ((pq[0][q] + pq[1][q]) + pq[2][q]) + pq[3][q];
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[4 * q + r] = t4[r];
          }
        }
        const int m = wave * 32 + lrow;
        const unsigned oy = __umulhi((unsigned)m, g.mP), ox = (unsigned)m - oy * (unsigned)P;
        const unsigned v = a + (unsigned)(j * R) - 1u + oy;                  // (a = 0, j = 0, oy = 0 wraps: fails v < e)
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
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[8 * cc + r4]), __float_as_uint(sum[8 * cc + 4 + r4]),
                                                             false, false);
            tv[r4] = __uint_as_float(sw[0]);
            tv[4 + r4] = __uint_as_float(sw[1]);
          }
          o.u = cx_pack8_stats(tv, valid, true, s1[cc], s2[cc]);
          if (valid && !((CX_PC_ABL & 2) && o.u.x != 0x12345u)) *reinterpret_cast<uint4*>(yrow + 8 * (2 * cc + lh)) = o.u;
        }
        PSTAMP(5)
      }
      bar_lds();                                               // the pass is over
#ifdef CX_PC_STAMPS
      if (tid == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 6; ++i) pc_stamps[blockIdx.x * 16 + i] += st_acc[i];
        pc_stamps[blockIdx.x * 16 + 7] += (unsigned long long)J;
      }
#endif
    }
  }

  if (stat_sum) {
    float* scratch = reinterpret_cast<float*>(part);             // the partial sums are no longer read
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
  const size_t smem = PART_BYTES + (size_t)(g.Q + 2) * XP;
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


#ifndef CX_PCW_ABL
#define CX_PCW_ABL 0     // timing ablations of diagnostic builds (results wrong): 1 no MFMAs, 2 fragment reads only in the first k-step, 4 no staging stores
#endif
// ================================================================================================= weight gradient
// dW[n][k][dy][dx] += sum over pixels of dY[px][n] * relu(bn(y1))[px + (dy-1, dx-1)][k]      (K = 128, N = 32; conv3x3_ring.hip's
// conv3x3_ring_wgrad_kernel and conv3x3_strip.hip hold the forms this one is measured against)
// The same roles and row space as the forward kernel above, and a better fit for them: there is no per-step epilogue (the 36
// accumulator tiles of a workgroup live in the consumers' registers until the end) and an MFMA needs ~1.1 KB of LDS fragments, not 2.
//   * PRODUCERS (waves 4-7): the R new rows of y1 (BN + ReLU while staged; ring of 2R + 2 rows, 320-byte pixels: conflict-free for
//     the transposing reads) and the R rows of dY of the NEXT step (two buffers of R x P pixels, 64-byte pixels); pad columns,
//     separator rows and rows outside the workgroup's range are zeros in the dY buffers, so whatever the y1 ring holds beside
//     them does not count;
//   * CONSUMERS (waves 0-3): wave w owns input channels [32w, 32w + 32) of all nine taps: nine 32 x 32 accumulators.  The
//     contraction runs over the flat padded pixels of the step in k-steps of 16: one dY^T fragment serves nine MFMAs, each with
//     the y1 fragment of its tap's shifted window, all by ds_read_b64_tr_b16 (pixel-major images, the pixel is the contraction
//     index); one barrier per step.
//   * at the end the accumulators meet in LDS in OIHW order and leave as one coalesced partial tile per workgroup (slab sums).
constexpr int AXP = 320;                 // y1 ring pitch: 128 bf16 + 64 B (four consecutive pixels of a transposing read: distinct banks)
constexpr int GXP = 64;                  // dY pitch: 32 bf16

// two transposing reads = the eight pixels of one MFMA operand.  The halves are joined by a vector shuffle + bit cast: assembled element
// by element (`r[0] = lo.e[0]; ...`, as the older kernels do) the compiler emits a v_bfi per dword on the loaded registers -- and an
// s_waitcnt lgkmcnt right behind the read it belongs to, i.e. every fragment read is waited for where it is ISSUED, not where it is used
__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  return cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1)));
}

template <int NCA, int NCG, int DEPTH, bool g_affine2>
__global__ __launch_bounds__(NT, 1) void conv3x3_pc_wgrad_kernel(const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2,
                                                                const float* __restrict__ ga, const float* __restrict__ gb,
                                                                const float* __restrict__ gc, const bf16* __restrict__ x, int ldx,
                                                                const float* __restrict__ pa, const float* __restrict__ pb,
                                                                float* __restrict__ dw, float* __restrict__ slab, const PcGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = g.P, R = g.R, NR = g.NR, Q = g.Q, W = g.W, H = g.H, Wt = g.Wt, Hs = g.Hs;
  const int nk = (R * P + 15) / 16;
  char* ring = smem;                                           // y1: [Q + 2][320 B]
  char* gbuf = smem + (size_t)(Q + 2) * AXP;                   // dY: [2][nk * 16][64 B]
  const int GB = nk * 16 * GXP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < ((Q + 2) * AXP + 2 * GB) / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const unsigned v0 = (unsigned)(((unsigned long long)blockIdx.x * g.V) / (unsigned)g.nwg);
  const unsigned v1 = (unsigned)(((unsigned long long)(blockIdx.x + 1) * g.V) / (unsigned)g.nwg);
  const unsigned nrows = v1 - v0;
  const unsigned rot = (g.rot && nrows > (unsigned)(2 * R)) ? (blockIdx.x * 37u) % nrows : 0u;
  auto pass_range = [&](int pass, unsigned& a, unsigned& e, int& J) __attribute__((always_inline)) {
    a = pass == 0 ? v0 + rot : v0;
    e = pass == 0 ? v1 : v0 + rot;
    J = e > a ? (int)((e - a + 1 + R - 1) / R) : 0;
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Barriers of a pass, the same J + 2 for both roles: one per step j = -1 .. J-1 (the rows of step j are staged), one at the end.
  if (wave >= 4) {
    // ================================================================================================= producers
    // (no raised priority here, unlike the forward kernel: this kernel is bound by the CONSUMERS' instruction stream -- 72 MFMAs, 160
    // transposing reads and their address arithmetic per step -- and the producers wait for them at the barrier half of the time;
    // A/B on one box: 234 us with the producers at priority 2, 222 at equal priority, see below for the consumers at 2)
    const int ptid = tid - NPT;
    const int c8 = ptid & 15, cg = ptid & 3;
    float csc[8], csh[8], qa[8], qb[8], qc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      csc[j] = pa[c8 * 8 + j]; csh[j] = pb[c8 * 8 + j];
      qa[j] = g_affine2 ? ga[cg * 8 + j] : 1.f; qb[j] = g_affine2 ? gb[cg * 8 + j] : 0.f; qc[j] = g_affine2 ? gc[cg * 8 + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(csc[j]), "v"(csh[j]), "v"(qa[j]), "v"(qb[j]), "v"(qc[j]));
    // y1 chunks: chunk id ptid + 256 i over R rows x P positions x 16 chunks; dY chunks: ... x 4 chunks (see the forward kernel: the
    // per-row part of an address is worked out once per step by lane r and fetched with ds_bpermute)
    const int cpa = P * 16, cpg = P * 4;
    int arow4[NCA], aoff[NCA], grow4[NCG], goff[NCG];
    uint32_t acof[NCA], gcof[NCG], gcof2[NCG];
    unsigned am0 = 0, am1 = 0, amrow = 0, gm0 = 0, gm1 = 0, gmrow = 0;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int cid = ptid + NPT * i;
      const int cr = cid / cpa, cp = (cid - cr * cpa) >> 4;
      arow4[i] = cr * 4;
      aoff[i] = cr * P + cp;
      acof[i] = ((uint32_t)cp * (uint32_t)ldx + (uint32_t)c8 * 8u) * 2u;
      amrow |= cr < R ? (1u << i) : 0u;
      am0 |= ((unsigned)(cp - 1) < (unsigned)W) ? (1u << i) : 0u;
      am1 |= ((unsigned)(Wt - 1 + cp) < (unsigned)W) ? (1u << i) : 0u;
    }
    am0 &= amrow; am1 &= amrow;
#pragma unroll
    for (int i = 0; i < NCG; ++i) {
      const int cid = ptid + NPT * i;
      const int cr = cid / cpg, cp = (cid - cr * cpg) >> 2;
      grow4[i] = cr * 4;
      goff[i] = cr * P + cp;
      gcof[i] = ((uint32_t)cp * (uint32_t)ldg + (uint32_t)cg * 8u) * 2u;
      gcof2[i] = ((uint32_t)cp * (uint32_t)ldg2 + (uint32_t)cg * 8u) * 2u;
      gmrow |= cr < R ? (1u << i) : 0u;
      // dY: only this tile's OWN columns (positions 1 .. Wt): the halo positions belong to the neighbouring tile's sums
      const bool own = cp >= 1 && cp <= Wt;
      gm0 |= (own && (unsigned)(cp - 1) < (unsigned)W) ? (1u << i) : 0u;
      gm1 |= (own && (unsigned)(Wt - 1 + cp) < (unsigned)W) ? (1u << i) : 0u;
    }
    gm0 &= gmrow; gm1 &= gmrow;
    const char* __restrict__ xb = reinterpret_cast<const char*>(x);
    const char* __restrict__ gsb = reinterpret_cast<const char*>(gsl);
    const char* __restrict__ g2b = reinterpret_cast<const char*>(g2);
    char* ringc = ring + c8 * 16;
    char* gbufc = gbuf + cg * 16;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      unsigned a, e;
      int J;
      pass_range(pass, a, e, J);
      if (J == 0) continue;
      uint4 pra[DEPTH][NCA], prg[DEPTH][NCG], prg2[DEPTH][g_affine2 ? NCG : 1];
      unsigned pma[DEPTH], pmg[DEPTH];
      // step j: y1 rows [a + jR, a + jR + R) and dY rows [a + jR - 1, a + jR + R - 1) (the rows step j multiplies) -> registers
      auto issue = [&](uint4 (&pa_)[NCA], uint4 (&pg_)[NCG], uint4 (&pg2_)[g_affine2 ? NCG : 1], unsigned& ma, unsigned& mg, int j) __attribute__((always_inline)) {
        const unsigned va = a + (unsigned)(j * R) + (unsigned)lane, vg = va - 1u;
        const unsigned ta = __umulhi(va, g.mHs), ya = va - ta * (unsigned)Hs;
        const unsigned tg = __umulhi(vg, g.mHs), yg = vg - tg * (unsigned)Hs;
        const unsigned live = (unsigned)(j < J) & (unsigned)(lane < R);
        const unsigned aok = live & (unsigned)(va < g.V) & (unsigned)(ya < (unsigned)H) & ((unsigned)(j >= 0) | (unsigned)(va + 2u >= a));
        const unsigned gok = live & (unsigned)(j >= 0) & (unsigned)(vg >= a) & (unsigned)(vg < e) & (unsigned)(yg < (unsigned)H);
        const unsigned txa = ta & (unsigned)(g.ntx - 1), txg = tg & (unsigned)(g.ntx - 1);
        const uint32_t pxa = ((ta >> g.ntx_shift) * (uint32_t)H + ya) * (uint32_t)W + txa * (uint32_t)Wt - 1u;
        const uint32_t pxg = ((tg >> g.ntx_shift) * (uint32_t)H + yg) * (uint32_t)W + txg * (uint32_t)Wt - 1u;
        const int worda = (int)((pxa * (uint32_t)ldx * 2u) | aok | (txa << 1));
        const int wordg = (int)((pxg << 2) | gok | (txg << 1));           // (pixel index: the two dY tensors have their own pitches)
        ma = 0; mg = 0;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute(arow4[i], worda);
          const unsigned ok = w & (((w & 2u) ? am1 : am0) >> i) & 1u;
          ma |= ok << i;
          pa_[i] = *reinterpret_cast<const uint4*>(xb + (size_t)(ok ? (w & ~15u) + acof[i] : 0u));
        }
#pragma unroll
        for (int i = 0; i < NCG; ++i) {
          const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute(grow4[i], wordg);
          const unsigned ok = w & (((w & 2u) ? gm1 : gm0) >> i) & 1u;
          mg |= ok << i;
          const uint32_t px = w >> 2;
          pg_[i] = *reinterpret_cast<const uint4*>(gsb + (size_t)(ok ? px * (uint32_t)ldg * 2u + gcof[i] : 0u));
          if constexpr (g_affine2) pg2_[i] = *reinterpret_cast<const uint4*>(g2b + (size_t)(ok ? px * (uint32_t)ldg2 * 2u + gcof2[i] : 0u));
        }
      };
      auto stage = [&](uint4 (&pa_)[NCA], uint4 (&pg_)[NCG], uint4 (&pg2_)[g_affine2 ? NCG : 1], unsigned ma, unsigned mg, int sb, int gsel) __attribute__((always_inline)) {
        const int sbp = sb * P, wrap_from = (NR - sb) * 4;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          uint4 o = cx_affine_relu8(pa_[i], csc, csh);
          const unsigned keep = 0u - ((ma >> i) & 1u);
          o.x &= keep; o.y &= keep; o.z &= keep; o.w &= keep;
          int pos = sbp + aoff[i] - (arow4[i] >= wrap_from ? Q : 0);
          pos = ((amrow >> i) & 1u) ? pos : Q + 1;                          // (chunks past the step's rows: the spare pixel, kept zero)
          if (!((amrow >> i) & 1u)) o = make_uint4(0, 0, 0, 0);
          if (!(CX_PCW_ABL & 4) || o.x == 0x12345u) *reinterpret_cast<uint4*>(ringc + pos * AXP) = o;
        }
        char* gdst = gbufc + gsel * GB;
#pragma unroll
        for (int i = 0; i < NCG; ++i) {
          uint4 o = pg_[i];
          if constexpr (g_affine2) o = cx_affine2_8(pg_[i], pg2_[i], qa, qb, qc);
          const unsigned keep = 0u - ((mg >> i) & 1u);
          o.x &= keep; o.y &= keep; o.z &= keep; o.w &= keep;
          if ((gmrow >> i) & 1u) *reinterpret_cast<uint4*>(gdst + goff[i] * GXP) = o;
        }
      };
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        issue(pra[d], prg[d], prg2[d], pma[d], pmg[d], -1 + d);
        __builtin_amdgcn_sched_barrier(0);
      }
      int sb = (2 * NR - R) % NR;                              // slot of y1 row a - R (the warm-up step's first row)
#ifdef CX_PC_STAMPS
      unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = pstamp();
#endif
      auto pstep = [&](int j, auto KI) __attribute__((always_inline)) {
        constexpr int k = decltype(KI)::value;
        PSTAMP(0)
#ifdef CX_PC_STAMPS
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NCA + NCG * (g_affine2 ? 2 : 1)) * (DEPTH - 1)) : "memory");
        PSTAMP(1)
#endif
        stage(pra[k], prg[k], prg2[k], pma[k], pmg[k], sb, j & 1);
        PSTAMP(2)
        issue(pra[k], prg[k], prg2[k], pma[k], pmg[k], j + DEPTH);
        sb += R;
        if (sb >= NR) sb -= NR;
        PSTAMP(3)
        bar_lds();                                             // barrier j: the rows of step j are staged
        PSTAMP(4)
      };
      for (int j = -1; j < J; j += DEPTH) {
        pstep(j, std::integral_constant<int, 0>());
        if (DEPTH > 1) { if (j + 1 >= J) break; pstep(j + 1, std::integral_constant<int, 1 % DEPTH>()); }
        if (DEPTH > 2) { if (j + 2 >= J) break; pstep(j + 2, std::integral_constant<int, 2 % DEPTH>()); }
      }
      bar_lds();                                               // the pass is over
#ifdef CX_PC_STAMPS
      if (tid == 256 && blockIdx.x < 1024) {
        for (int i = 0; i < 5; ++i) pc_stamps[blockIdx.x * 16 + 8 + i] += st_acc[i];
        pc_stamps[blockIdx.x * 16 + 15] += (unsigned long long)(J + 1);
      }
#endif
    }
  } else {
    // ================================================================================================= consumers
    // transposing reads: lane 4q + p of a 16-lane group supplies row q (then q + 4) of its block, columns 4p .. 4p + 3; group gq: rows
    // 8 (gq >> 1) + .., columns 16 (gq & 1) + ..  -> this lane's pixel inside a k-step and its byte offset inside a pixel
#ifndef CX_PCW_CPRIO
#define CX_PCW_CPRIO 2
#endif
    __builtin_amdgcn_s_setprio(CX_PCW_CPRIO);
    const int gq = lane >> 4, rq = (lane & 15) >> 2, pp = lane & 3;
    const int lpx = 8 * (gq >> 1) + rq;                        // second read: + 4
    const int acol = (wave * 32 + 16 * (gq & 1) + 4 * pp) * 2; // y1 channels of this wave
    const int gcol = (16 * (gq & 1) + 4 * pp) * 2;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      unsigned a, e;
      int J;
      pass_range(pass, a, e, J);
      if (J == 0) continue;
      int sb = (2 * NR - R - 2) % NR;                          // slot of y1 row a + jR - 2 for j = -1
#ifdef CX_PC_STAMPS
      unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = pstamp();
#endif
      bar_lds();                                               // barrier -1 (warm-up rows)
      for (int j = 0; j < J; ++j) {
        sb += R;
        if (sb >= NR) sb -= NR;
        PSTAMP(0)
        bar_lds();                                             // barrier j
        PSTAMP(1)
        const char* gcur = gbuf + (j & 1) * GB + gcol;
        // flat ring index of (dY pixel m, tap (dy, dx)): sb*P + m + dy*P + dx - 1 (mod Q; dx may run into the two spare pixels: zero)
        const int wsm1 = sb * P - 1 + Q;
        // The 3 nk groups (k-step, dy) of three MFMAs run as one software pipeline: the fragment reads of a group go out TWO groups
        // (six MFMAs, ~190 cycles) ahead, across the k-step boundary -- one group ahead (the first form of this loop) every group
        // waited out the LDS latency: 126 cycles per MFMA (profiles/r05_pc_wgrad.txt).  Register set = dy.
        // byte offsets (ring-relative, this lane's column included) of the six rows a k-step's y1 fragments start from: (dy, half h).
        // They advance by 16 pixels per k-step with ONE conditional wrap -- computed from scratch per k-step (two wraps and a
        // multiply per address) the consumer issued ~100 vector instructions per nine MFMAs and was bound by them
        int ab[3][2];
        const int QA = Q * AXP, lim = QA + acol;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int f = wsm1 + lpx + 4 * h + dy * P;
            if (f >= Q) f -= Q;
            if (f >= Q) f -= Q;
            ab[dy][h] = (int)__umul24((unsigned)f, (unsigned)AXP) + acol;
          }
        auto addr = [&](int) __attribute__((always_inline)) {     // the next k-step
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int t = ab[dy][h] + 16 * AXP;
              ab[dy][h] = t >= lim ? t - QA : t;
            }
        };
        bf16x8 af[3][3], gf, gfn;
        bool rd_on = true;
        auto rd = [&](int dy) __attribute__((always_inline)) {
          if ((CX_PCW_ABL & 2) && !rd_on) return;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) af[dy][dx] = tr_pair(ring + ab[dy][0] + dx * AXP, ring + ab[dy][1] + dx * AXP);
        };
#ifndef CX_PCW_SCHED
#define CX_PCW_SCHED 1
#endif
        auto mm = [&](int dy) __attribute__((always_inline)) {
          if (CX_PCW_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            if (CX_PCW_ABL & 1) { acc[dy * 3 + dx][0] += (float)af[dy][dx][0] + (float)gf[0]; continue; }
            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, af[dy][dx], acc[dy * 3 + dx], 0, 0, 0);   // D[row = n][col = k]
            // an MFMA holds the vector issue for 8 of its 32 cycles: the address arithmetic of the next k-step and the fragment reads
            // issue in the gaps (one MFMA, then up to four VALU and two LDS instructions), not in a block between the MFMA triples
            if (CX_PCW_SCHED == 1) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS read
            }
          }
          if (CX_PCW_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
        };
        gf = tr_pair(gcur + lpx * GXP, gcur + (lpx + 4) * GXP);
        rd(0);
        rd(1);
        for (int ks = 0; ks < nk; ++ks) {
          rd(2);                                               // (ks, 2), with this k-step's addresses
          rd_on = false;
          const int kn = ks + 1 < nk ? ks + 1 : ks;            // (the last k-step re-reads itself: no branch in the pipeline)
          addr(kn);
          mm(0);
          gfn = tr_pair(gcur + (kn * 16 + lpx) * GXP, gcur + (kn * 16 + lpx + 4) * GXP);
          rd(0);                                               // (ks + 1, 0)
          mm(1);
          rd(1);                                               // (ks + 1, 1)
          mm(2);
          gf = gfn;
        }
#ifdef CX_PC_STAMPS
        asm volatile("" ::"v"(acc[8][0]));
#endif
        PSTAMP(2)
      }
      bar_lds();                                               // the pass is over
#ifdef CX_PC_STAMPS
      if (tid == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 3; ++i) pc_stamps[blockIdx.x * 16 + i] += st_acc[i];
        pc_stamps[blockIdx.x * 16 + 7] += (unsigned long long)J;
      }
#endif
    }
  }

  // ---- the workgroup's partial dW in OIHW order through LDS (the rings are free), then one coalesced tile
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                 // [32 n][128 k][9]
  if (wave < 4) {
    const int lrow = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;
        red[(n * 128 + wave * 32 + lrow) * 9 + t] = acc[t][r];
      }
  }
  __syncthreads();
  for (int idx = tid; idx < 32 * 128 * 9; idx += NT)
    dw_out(dw, slab, (size_t)32 * 128 * 9, (int)blockIdx.x, (size_t)idx, red[idx]);
}

template <int NCA, int NCG, int DEPTH, bool A2>
int launch_pc_wgrad_a(const CxWgrad& p, hipStream_t st, const PcGeo& g, size_t smem) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pc_wgrad_kernel<NCA, NCG, DEPTH, A2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const size_t wtotal = (size_t)32 * 128 * 9;
  float* slab = dw_slab(p.scratch, p.scratch_floats, g.nwg, (long long)wtotal);
  CX_KTAG("conv3x3_pc_wgrad_kernel<%d, %d, %d, %s>", NCA, NCG, DEPTH, A2 ? "true" : "false");
  hipLaunchKernelGGL((conv3x3_pc_wgrad_kernel<NCA, NCG, DEPTH, A2>), dim3(g.nwg), dim3(NT), smem, st, (const bf16*)p.g, p.ldg,
                     (const bf16*)(A2 ? p.g2 : p.g), A2 ? p.ldg2 : p.ldg, p.ga, p.gb, p.gc, (const bf16*)p.x, p.ldx, p.pa, p.pb, p.dw, slab, g);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, wtotal, g.nwg, st) : 0;
}
template <int NCA, int NCG, int DEPTH>
int launch_pc_wgrad(const CxWgrad& p, hipStream_t st, const PcGeo& g, size_t smem) {
  return p.g_prologue == CX_PRO_AFFINE2 ? launch_pc_wgrad_a<NCA, NCG, DEPTH, true>(p, st, g, smem)
                                        : launch_pc_wgrad_a<NCA, NCG, DEPTH, false>(p, st, g, smem);
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

// Eligibility + launch, called from cx_conv_gemm ahead of the ring kernel -- ONLY when CxConv.kernel_hint asks for it (form 8): measured
// on MI355X it ties the ring kernel on 80x80 / 40x40 maps and loses on smaller ones (profiles/r05_pc_fwd.txt: stamps and timing
// ablations; the SIMD's vector issue -- ~45 instructions per staged 16-byte chunk beside the MFMAs -- is as long as the HBM time of a
// step, whichever wave issues them).  It stays in the library as the measured alternative with its tests, not as a default.
int cx_try_pc_fwd(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (((p.kernel_hint >> 8) & 0xff) != 9) return 0;
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
  return launch_pc<8, 3>(p, st, g);
}

// The dense layers' 3x3 weight gradient on producer / consumer waves; called from cx_conv_wgrad ahead of the ring / strip kernels.
// CxWgrad.kernel_hint form 8 selects it (tests, A/B); form 7 pins the ring / strip kernels.
// Like the forward kernel it is the MEASURED ALTERNATIVE, taken only when CxWgrad.kernel_hint asks for it (form 8): 222 us against
// 219 for the ring kernel on 80x80 maps at bs = 256 (116 against 120 at bs = 128), 73 against 69 on 40x40, the DenseNet121 step 26.83 ms
// with it against 26.87 without (three alternating runs on one box) -- profiles/r05_pc_wgrad.txt: the consumers' own stream
// (72 MFMAs + 160 transposing reads + their addresses per step) takes ~85 cycles per MFMA whatever its order or priority.
#ifndef CX_PCW_DEFAULT
#define CX_PCW_DEFAULT 0
#endif
int cx_try_pc_wgrad(const CxWgrad& p, hipStream_t st, bool* handled) {
  *handled = false;
  const int form = ((p.kernel_hint >> 8) & 0xff) - 1;          // 7: the ring / strip kernels, 8: this kernel
  if (form == 7 || (!CX_PCW_DEFAULT && form != 8)) return 0;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1 || p.dil > 1) return 0;
  if (p.K != 128 || p.N != 32 || p.x_prologue != CX_PRO_AFFINE_RELU || p.dtype != CX_DT_BF16) return 0;
  if (p.g_prologue != CX_PRO_NONE && p.g_prologue != CX_PRO_AFFINE2) return 0;
  if (p.W < 4 || p.H < 1) return 0;
  {
    unsigned long long ldm = (unsigned long long)(p.ldx > p.ldg ? p.ldx : p.ldg);
    if (p.g_prologue == CX_PRO_AFFINE2 && (unsigned long long)p.ldg2 > ldm) ldm = (unsigned long long)p.ldg2;
    if ((unsigned long long)p.B * p.H * p.W * ldm * 2ull >= (1ull << 32)) return 0;      // 32-bit chunk offsets in the kernel
    if ((unsigned long long)p.B * p.H * p.W >= (1ull << 30)) return 0;                    // pixel index in the per-row word
  }
  PcGeo g;
  g.H = p.H; g.W = p.W;
  g.ntx = (p.W >= 64 && p.W % 2 == 0) ? 2 : 1;
  g.ntx_shift = g.ntx - 1;
  g.Wt = p.W / g.ntx;
  g.P = g.Wt + 2;
  g.Hs = p.H + 1;
  // largest R with <= 8 y1 chunks / 2 dY chunks per producer thread (R x P <= 128 flat pixels = eight k-steps) and both images in LDS
  int R = (8 * NPT) / (g.P * 16);
  for (; R >= 1; --R) {
    const int nk = (R * g.P + 15) / 16;
    if ((size_t)((2 * R + 2) * g.P + 2) * AXP + (size_t)2 * nk * 16 * GXP <= 160 * 1024 && R * g.P * 4 <= 2 * NPT) break;
  }
  if (R < 1) return 0;
  if (R > g.Hs) R = g.Hs;
  g.R = R; g.NR = 2 * R + 2; g.Q = g.NR * g.P;
  const unsigned long long V = (unsigned long long)p.B * g.ntx * g.Hs;
  if (V * (unsigned long long)g.Hs >= (1ull << 32) || V + 4096 >= (1ull << 31)) return 0;
  g.V = (unsigned)V;
  g.mP = 0xffffffffu / (unsigned)g.P + 1u;
  g.mHs = 0xffffffffu / (unsigned)g.Hs + 1u;
  int nwg = 256;
  if (V < (unsigned long long)nwg * 2 * R) nwg = (int)((V + 2 * R - 1) / (2 * R));
  if (nwg < 1) nwg = 1;
  g.nwg = nwg;
  g.rot = 1;
  const int nk = (R * g.P + 15) / 16;
  const size_t smem0 = (size_t)(g.Q + 2) * AXP + (size_t)2 * nk * 16 * GXP;
  const size_t smem = smem0 < (size_t)32 * 128 * 9 * 4 ? (size_t)32 * 128 * 9 * 4 : smem0;       // (the OIHW tile at the end)
  *handled = true;
  const int needa = (R * g.P * 16 + NPT - 1) / NPT;
  if (needa <= 4) return launch_pc_wgrad<4, 1, 3>(p, st, g, smem);
  if (needa <= 6) return launch_pc_wgrad<6, 2, 2>(p, st, g, smem);
  return launch_pc_wgrad<8, 2, 2>(p, st, g, smem);
}
