// NOT part of the library: the 8-frame / ring-of-four form of the bf16 pair stage tried in round 3 (kernel body only; it
// was compiled inside csrc/tspn_bf16.hip, whose helpers -- glds16, relu_pack, HP_KC, the vector typedefs -- it uses).
// Bit-identical results, 3.32 ms against 3.10 ms for the shipped 16-frame form at cfg3 (4 videos): the two object-parity
// half-groups of a column pair read the SAME 16 bytes of a subject row and ds_read_b128 does not broadcast
// (SQ_LDS_BANK_CONFLICT 118 M per launch against 0), waves parked 44 % against 34 %.
// See profiles/r3/bf16_pair_stage_counters.md.
// ------------------------------------------------------------------------------------------------
// Pair stage, bf16, second form (round 3) for N > 12: the same arithmetic, operation for operation, as
// heads_pairgrid_bf16_kernel -- bit-identical results -- on a tile of 16 subjects x 16 objects x EIGHT frames.
// The counters of the 16-frame form (profiles/r3/bf16_pair_stage_counters.md) say a wave is parked 34 % of its life at
// the end-of-k-step vmcnt(0) + barrier: 64 KB per k-step at one k-step of DMA lookahead is all the LDS holds.  An MFMA
// column does not have to be a frame: here the 16 columns of v_mfma_f32_16x16x32_bf16 are 8 frames x 2 OBJECTS, so a
// k-step stages 32 rows x 8 frames x 32 channels = 32 KB (+ 1 KB of head weights) and the LDS holds a RING OF FOUR
// stages, filled three k-steps ahead; the wait at the top of a k-step is for pieces requested three k-steps ago
// (in-order VMEM return: vmcnt(2 x pieces per stage)), one bare s_barrier per k-step.
//   wave (ws, wo) = subjects 4 ws .. 4 ws + 3 x objects 8 wo .. 8 wo + 7 (four object pairs): 16 MFMAs per k-step;
//   lane = (column = (frame f8 = l & 7, object parity oj = (l >> 3) & 1), channel group kg = l >> 4);
//   LDS row image (1 KB = 8 frames x 8 quads of 16 bytes): position 16 X + slot, X = 2 (q >> 2) + (q & 1),
//   slot = f8 + 8 (((q >> 1) & 1) ^ (row & 1)) -- the row-parity swizzle makes the two object rows of a column pair
//   land in complementary halves of the 16 slots, so the 16 lanes of a ds_read_b128 group cover all 64 banks; one DMA
//   piece = one row of one k-step = 8 complete 128-byte lines of y.
constexpr int H2_FB = 8;
constexpr int H2_ROWB = H2_FB * HP_KC * 4;        // 1024 B per row and k-step
constexpr int H2_ROWS = 32;                       // 16 subjects + 16 objects
constexpr int H2_ST = H2_ROWS * H2_ROWB + 1024;   // + the k-step's slice of the head weights
constexpr int H2_NST = 4;

template <int VM>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
}

__global__ __launch_bounds__(512, 1) void heads_pairgrid_bf16_f8_kernel(
    const float* __restrict__ y, int64_t ldm, int B, int N, int C, int T,
    const __bf16* __restrict__ Whp, const float* __restrict__ bh, int H, float* __restrict__ out,
    int nsb, int nob, int nfb) {
  constexpr int SW = 4, OW = 8, OP = OW / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ob = wg % nob;
  wg /= nob;
  const int sb = wg % nsb;
  wg /= nsb;
  const int fb = wg % nfb;
  const int b = wg / nfb;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ws = wave & 3, wo = wave >> 2;
  const int f8 = lane & 7, oj = (lane >> 3) & 1, kg = lane >> 4;
  const int t0 = fb * H2_FB;

  // DMA sources: wave w stages rows 4 w .. 4 w + 3 (one piece each).  Lane l of a piece lands at position l of the
  // row image: X = l >> 4, slot = l & 15 -> frame slot & 7, quad 4 (X >> 1) + 2 ((slot >> 3) ^ (row & 1)) + (X & 1)
  const float* src[4];
  {
    const int X = lane >> 4, slot = lane & 15;
    const int fr = slot & 7, hb = slot >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 4 + i;                    // row parity = i & 1
      const int q = 4 * (X >> 1) + 2 * (hb ^ (i & 1)) + (X & 1);
      int trk = r < 16 ? sb * 16 + r : ob * 16 + r - 16;
      trk = min(trk, N - 1);
      const int t = min(t0 + fr, T - 1);
      src[i] = y + (((int64_t)b * N + trk) * T + t) * ldm + (r < 16 ? 0 : C) + 4 * q;
    }
  }
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(Whp) + lane;          // + 64 per k-step
  auto stage = [&](int slot_i) {
    char* dst = smem + slot_i * H2_ST + wave * 4 * H2_ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(src[i], dst + i * H2_ROWB);
      src[i] += HP_KC;
    }
    if (wave == 0) {
      glds16(wsrc, smem + slot_i * H2_ST + H2_ROWS * H2_ROWB);
      wsrc += 64;
    }
  };

  f32x4 acc[SW][OP];
#pragma unroll
  for (int s = 0; s < SW; ++s)
#pragma unroll
    for (int o = 0; o < OP; ++o) acc[s][o] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = C / HP_KC;
  // fragment offsets of lane (f8, oj, kg) inside a row image: quad 2 kg at X0 = 2 (kg >> 1), quad 2 kg + 1 256 bytes on
  const int x0 = 2 * (kg >> 1);
  const int uoff[2] = {(16 * x0 + f8 + 8 * (kg & 1)) * 16, (16 * x0 + f8 + 8 * ((kg & 1) ^ 1)) * 16};   // row parity 0 / 1
  const int voff = (16 * x0 + f8 + 8 * ((kg & 1) ^ oj)) * 16 + oj * H2_ROWB;          // object 2 op + oj of a pair
  // prologue: stages 0 .. 2
  for (int i = 0; i < H2_NST - 1 && i < nk; ++i) stage(i);

  for (int k = 0; k < nk; ++k) {
    // stage k has landed: everything but the pieces of the (up to two) younger stages
    const int younger = min(nk - 1 - k, H2_NST - 2);
    if (wave == 0) {
      if (younger == 2) wait_vm<10>(); else if (younger == 1) wait_vm<5>(); else wait_vm<0>();
    } else {
      if (younger == 2) wait_vm<8>(); else if (younger == 1) wait_vm<4>(); else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();       // every wave's pieces of stage k are in; everybody has left stage k - 1
    if (k + H2_NST - 1 < nk) stage((k + H2_NST - 1) % H2_NST);
    __builtin_amdgcn_sched_barrier(0);
    const char* base = smem + (k % H2_NST) * H2_ST;
    const bf16x8 wfrag = *reinterpret_cast<const bf16x8*>(base + H2_ROWS * H2_ROWB + lane * 16);
    f32x4 u[SW][2];
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      u[s][0] = *reinterpret_cast<const f32x4*>(base + (SW * ws + s) * H2_ROWB + uoff[s & 1]);
      u[s][1] = *reinterpret_cast<const f32x4*>(base + (SW * ws + s) * H2_ROWB + uoff[s & 1] + 256);
    }
    const char* vbase = base + (16 + OW * wo) * H2_ROWB + voff;
#pragma unroll
    for (int o = 0; o < OP; ++o) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(vbase + 2 * o * H2_ROWB);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(vbase + 2 * o * H2_ROWB + 256);
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        const f32x4 a0 = u[s][0] + v0, a1 = u[s][1] + v1;
        u32x4 pk = {relu_pack(a0[0], a0[1]), relu_pack(a0[2], a0[3]), relu_pack(a1[0], a1[1]),
                    relu_pack(a1[2], a1[3])};
        acc[s][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag, __builtin_bit_cast(bf16x8, pk),
                                                             acc[s][o], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this k-step's LDS reads have returned before its stage is refilled
  }

  // epilogue: lane = (frame f8, object parity oj, head group hg): heads 4 hg .. 4 hg + 3
  const int t = t0 + f8;
  const int hg = lane >> 4;
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = (4 * hg + r < H) ? bh[4 * hg + r] : 0.f;
#pragma unroll
  for (int s = 0; s < SW; ++s) {
    const int sg = sb * 16 + SW * ws + s;
#pragma unroll
    for (int o = 0; o < OP; ++o) {
      const int og = ob * 16 + OW * wo + 2 * o + oj;
      if (sg >= N || og >= N || sg == og || t >= T) continue;
      const int64_t p = (int64_t)b * N * (N - 1) + (int64_t)sg * (N - 1) + (og < sg ? og : og - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 4 * hg + r;
        if (h < H) out[(p * H + h) * T + t] = acc[s][o][r] + bias[r];
      }
    }
  }
}

