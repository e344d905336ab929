// REJECTED FORM of the bf16 pair stage (round 3), kept for the record; build it in place of csrc/tspn_bf16.hip with
//     TSPN_VARIANT_SRC=<a copy of csrc/tspn_bf16.hip with the kernel below pasted over heads_pairgrid_bf16_kernel's
//     head and loop, and smem = 4 * (sblk * HP_ROW + 1024) in the launcher> tools/build_variant.sh ring4 tspn_bf16.hip
// Idea: only the V rows through LDS (33 KB per stage, a ring of FOUR filled three k-steps ahead); the U fragments of
// a wave's own subjects straight into registers two k-steps ahead (three register sets in rotation -- a copy between
// the load and its wait lets the compiler read the registers before the wait); U loads issued before the DMA of the
// same k-step so that one counted vmcnt leaves exactly the youngest DMA in flight.  Bit-identical results.
// Measured (tools/time_hpb.py 4 7): 3.23 - 3.26 ms against 3.14 - 3.22 ms for the shipped two-stage form; SQ counters
// unchanged (waves waiting 33 % of their life, VALU issuing 53 %): the time a wave is parked at the end of a k-step is
// NOT the latency of the operand DMA -- with eight waves of identical work per workgroup and two per SIMD it is the
// wait for the other wave of the SIMD (the packed fp32 adds occupy the VALU 8 cycles each: the pipe is ~68 % busy).
// NW waves; workgroup = 2 NW subjects x OB objects x 16 frames, wave w owns SW subjects x OW objects.
// <4, 8>: 8 x 8 pairs, 2 workgroups/CU (small N).  <8, 16>: 16 x 16 pairs, 1 workgroup/CU -- half the bytes streamed
// from L2 per activation.
// Operands (round 3, second form): only the V rows (objects) go through LDS -- 32 KB + the k-step's head weights per
// stage at 16 objects, a RING OF FOUR stages filled THREE k-steps ahead; the U rows of a wave's own subjects are not
// shared with any other wave, so each lane fetches its fragment (8 channels of its frame = 32 contiguous bytes, the
// four lanes of a frame one 128-byte line) straight into registers one k-step ahead.  The first form staged U and V
// together, 64 KB per stage: two stages were all the LDS held, one k-step of lookahead against a DMA latency of about
// one k-step under load, and the waves sat at the end-of-k-step wait 34 % of their life (3.2 ms against a 2.0 ms issue
// floor, profiles/r3/bf16_pair_stage_counters.md).  VMEM operations retire in order: per k-step the U loads of k + 1 are
// issued BEFORE the DMA of k + 3, so the counted wait for them leaves exactly that DMA in flight.
template <int NW, int OB, int SW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void heads_pairgrid_bf16_kernel(
    const float* __restrict__ y, int64_t ldm, int B, int N, int C, int T,
    const __bf16* __restrict__ Whp, const float* __restrict__ bh, int H, float* __restrict__ out,
    int nsb, int nob, int nfb) {
  constexpr int SBLK = 2 * NW;
  constexpr int WS = SBLK / SW, WO = NW / WS, OW = OB / WO;   // waves along subjects / objects, objects per wave
  static_assert(WS * WO == NW && OW * WO == OB, "wave tiling");
  constexpr int ST = OB * HP_ROW + 1024;    // V rows + the k-step's slice of the head weights (one piece)
  constexpr int NSTG = 4, AHEAD = NSTG - 1;
  constexpr int VP = 2 * OB / NW;           // DMA pieces per wave and stage (two per row)
  static_assert(VP * NW == 2 * OB && VP == 4, "each wave stages two V rows");
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
  const int f = lane & 15, kg = lane >> 4;
  const int t0 = fb * HP_FB;
  const int ws = wave % WS, wo = wave / WS;      // this wave's subjects SW ws .., objects OW wo ..

  // DMA sources: wave w stages V rows 2w, 2w + 1 (2 pieces each) = objects OB ob + 2w .. (channels [C, 2C))
  // LDS image of a row: two pieces of 8 frames; inside a piece position = 16 X + slot with
  //   slot = (f & 7) + 8 ((q >> 1) & 1),  X = 2 (q >> 2) + (q & 1)      (q = 16-byte channel quad 0..7)
  // so that (a) one DMA piece fetches 8 complete 128-byte lines of y (8 frames x 32 channels) and
  // (b) the fragment read of lane (f, kg) for quad 2 kg + r sits at slot (f & 7) + 8 (kg & 1): the
  // four 16-lane groups of a ds_read_b128 each cover all 16 slots -- conflict-free.
  const float* src[VP];
  {
    const int fq = lane & 7;
    const int q = (lane >> 5) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1);
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int r = wave * 2 + (i >> 1), j = i & 1;
      const int trk = min(ob * OB + r, N - 1);
      const int t = min(t0 + 8 * j + fq, T - 1);
      src[i] = y + (((int64_t)b * N + trk) * T + t) * ldm + C + 4 * q;
    }
  }
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(Whp) + lane;          // + 64 per k-step
  auto stage = [&](int buf) {
#if defined(TSPN_HPB_ABL_NODMA)
    return;
#endif
    char* dst = smem + buf * ST + wave * 2 * HP_ROW;
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      glds16(src[i], dst + i * 1024);
      src[i] += HP_KC;
    }
    // head weights of the k-step, [4 kg][16 h][8 ch] bf16 = the packed layout itself, through LDS as well
    if (wave == 0) {
      glds16(wsrc, smem + buf * ST + OB * HP_ROW);
      wsrc += 64;
    }
  };
  // U fragments of the lane: subjects SW ws + s, frame t0 + f, channels 32 k + 8 kg .. + 7 (two 16-byte loads)
  const char* usrc[SW];
#pragma unroll
  for (int s = 0; s < SW; ++s) {
    const int trk = min(sb * SBLK + SW * ws + s, N - 1);
    usrc[s] = reinterpret_cast<const char*>(y + (((int64_t)b * N + trk) * T + min(t0 + f, T - 1)) * ldm + 8 * kg);
  }
  auto load_u = [&](f32x4 (&u)[SW][2]) {
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(u[s][0]) : "v"(usrc[s]) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(u[s][1]) : "v"(usrc[s]) : "memory");
      usrc[s] += HP_KC * 4;
    }
  };

  f32x4 acc[SW][OW];
#pragma unroll
  for (int s = 0; s < SW; ++s)
#pragma unroll
    for (int o = 0; o < OW; ++o) acc[s][o] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = C / HP_KC;
  // fragment of lane (f, kg): quads 2 kg (here) and 2 kg + 1 (256 bytes further)
  const int frag_off = (64 * (f >> 3) + 32 * (kg >> 1) + 8 * (kg & 1) + (f & 7)) * 16;
  f32x4 ua[SW][2], ub[SW][2], uc[SW][2];              // three sets in rotation: no register copies between a load and its wait
  load_u(ua);                                         // U(0), U(1): the OLDEST operations, any counted wait below covers them
  if (nk > 1) load_u(ub);
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < nk) stage(i);
  // U(0), U(1) and V(0) have landed once at most the pieces of V(1), V(2) are outstanding
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto kstep = [&](int k, f32x4 (&u)[SW][2], f32x4 (&un)[SW][2], f32x4 (&ul)[SW][2]) {
    const int buf = k & (NSTG - 1);
    if (k + 2 < nk) load_u(ul);                       // U(k + 2), BEFORE the DMA below: see the wait at the end of the k-step
    if (k + AHEAD < nk) stage((k + AHEAD) & (NSTG - 1));   // the stage k-step k - 1 has left (barrier at its end)
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 wfrag = *reinterpret_cast<const bf16x8*>(smem + buf * ST + OB * HP_ROW + lane * 16);
    const char* base = smem + buf * ST + frag_off;
    // V fragments are read two objects ahead of their use (LDS latency off the critical path).  The
    // reads and their counted waits are written out: left to itself the compiler issues every
    // fragment read right before its first use and waits for it at once (32 exposed LDS round trips
    // per k-step).  LDS returns in order, so "lgkmcnt(n)" = all but the newest n reads have landed.
    const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(base + (OW * wo) * HP_ROW);
    f32x4 vq[3][2];
#define TSPN_VREAD(slot, o)                                                                             \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[slot][0]) : "v"(vaddr), "n"((o) * HP_ROW)); \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[slot][1]) : "v"(vaddr), "n"((o) * HP_ROW + 256));
    TSPN_VREAD(0, 0)
    TSPN_VREAD(1, 1)
#pragma unroll
    for (int o = 0; o < OW; ++o) {
      if (o + 2 < OW) {
        TSPN_VREAD((o + 2) % 3, o + 2)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      } else if (o + 1 < OW) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vq[o % 3][0]), "+v"(vq[o % 3][1]));
      }
      const f32x4 v0 = vq[o % 3][0], v1 = vq[o % 3][1];
#pragma unroll
      for (int s = 0; s < SW; ++s) {
#if defined(TSPN_HPB_ABL_NOVALU)
        u32x4 pk = {__builtin_bit_cast(unsigned, u[s][0][0]) ^ __builtin_bit_cast(unsigned, v0[0]),
                    __builtin_bit_cast(unsigned, u[s][0][1]) ^ __builtin_bit_cast(unsigned, v0[2]),
                    __builtin_bit_cast(unsigned, u[s][1][0]) ^ __builtin_bit_cast(unsigned, v1[1]),
                    __builtin_bit_cast(unsigned, u[s][1][1]) ^ __builtin_bit_cast(unsigned, v1[3])};
#else
        const f32x4 a0 = u[s][0] + v0, a1 = u[s][1] + v1;
        u32x4 pk = {relu_pack(a0[0], a0[1]), relu_pack(a0[2], a0[3]), relu_pack(a1[0], a1[1]),
                    relu_pack(a1[2], a1[3])};
#endif
        acc[s][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag, __builtin_bit_cast(bf16x8, pk),
                                                             acc[s][o], 0, 0, 0);
      }
    }
#undef TSPN_VREAD
    // U(k + 1) -- and everything older: V(k + 1), V(k + 2) -- has landed when at most the pieces of V(k + 3) issued
    // after it are outstanding (wave 0 issues one piece more: its count is one operation stricter than needed)
    // needed at the next k-step: U(k + 1) (issued one k-step ago) and V(k + 1) (older).  Younger than U(k + 1): the
    // pieces of V(k + 2), U(k + 2), the pieces of V(k + 3) -- wave 0 issues one piece more per stage: stricter, harmless
    if (k + AHEAD < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * VP + 2 * SW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < SW; ++s) asm volatile("" : "+v"(un[s][0]), "+v"(un[s][1]));   // their uses stay below the wait
    __builtin_amdgcn_s_barrier();
  };
  for (int k = 0; k < nk; k += 3) {
    kstep(k, ua, ub, uc);
    if (k + 1 < nk) kstep(k + 1, ub, uc, ua);
    if (k + 2 < nk) kstep(k + 2, uc, ua, ub);
  }

