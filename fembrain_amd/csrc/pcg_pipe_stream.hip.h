// The streamed part of a row product of k_pcg_pipe, written in gfx950 assembly: y += sum over the slots [first, first + n) of a
// SELL slice of (3x3 fp32 block) x (gathered fp64 3-vector), lane = row.
//
// Why assembly.  The loop is the memory stream of the solver (9 x 256-byte value segments + one column word + three gathered
// doubles per slot and wavefront) and is only as fast as the number of loads it keeps in flight.  With the 42 registers of
// solver state per lane that the pipelined recurrences keep live, hipcc (ROCm 7.2) at the 168-register budget of a 12-wavefront
// workgroup serialises the nine value loads of a slot through ONE register with an s_waitcnt vmcnt(0) after each -- the same
// source at a 256-register budget gets 17 loads in flight in 159 registers.  Here the pipeline is fixed by hand:
//   C(k+2) column word of slot k+2          1 load      issued two slots ahead (the gathers depend on it)
//   V(k+1) the nine values of slot k+1      9 loads     issued one slot ahead
//   G(k+1) the three gathers of slot k+1    3 loads     issued as soon as C(k+1) has landed
//   M(k)   27 fp64 operations of slot k                 when V(k) and G(k) have landed
// Vector memory loads return in order, so "C(k+1) has landed" is s_waitcnt vmcnt(22) (V(k) 9 + G(k) 3 + C(k+2) 1 + V(k+1) 9
// younger loads may still fly) and "V(k), G(k) have landed" is vmcnt(13): 13..25 loads are in flight all the time.  The last two
// slots run with their own counts (nothing younger exists), so no load is ever issued past the slice.  Register sets alternate
// by slot parity (A / B).  Arithmetic: exactly the compiler's sequence for
//     y_a += (double)v[3a] * x0 + (double)v[3a+1] * x1 + (double)v[3a+2] * x2        (-ffp-contract=off: mul, mul, add, mul, add, add)
// so a product has the bits of k_spmv's.
// Registers: operands are allocated by the compiler; the 36 temporaries are v120..v155 (clobbers) -- a kernel using this has at
// least 156 registers per lane, which every k_pcg_pipe instantiation needs for its state anyway.
#pragma once
#include <hip/hip_runtime.h>

namespace fb {

// clang-format off
#define FBP_LOADV(r0, r1, r2, r3, r4, r5, r6, r7, r8)                      \
  "global_load_dword " r0 ", %[voff], %[svals]\n\t"                        \
  "global_load_dword " r1 ", %[voff], %[svals] offset:256\n\t"             \
  "global_load_dword " r2 ", %[voff], %[svals] offset:512\n\t"             \
  "global_load_dword " r3 ", %[voff], %[svals] offset:768\n\t"             \
  "global_load_dword " r4 ", %[voff], %[svals] offset:1024\n\t"            \
  "global_load_dword " r5 ", %[voff], %[svals] offset:1280\n\t"            \
  "global_load_dword " r6 ", %[voff], %[svals] offset:1536\n\t"            \
  "global_load_dword " r7 ", %[voff], %[svals] offset:1792\n\t"            \
  "global_load_dword " r8 ", %[voff], %[svals] offset:2048\n\t"            \
  "v_add_u32_e32 %[voff], 0x900, %[voff]\n\t"
#define FBP_LOADV_A FBP_LOADV("v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144")
#define FBP_LOADV_B FBP_LOADV("v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153")
// column word: 16-bit difference to the row (C16) or the 32-bit column
#define FBP_LOADC16(c) "global_load_sshort " c ", %[coff], %[scols]\n\t" "v_add_u32_e32 %[coff], 0x80, %[coff]\n\t"
#define FBP_LOADC32(c) "global_load_dword " c ", %[coff], %[scols]\n\t" "v_add_u32_e32 %[coff], 0x100, %[coff]\n\t"
// byte offset of the column in a plane (v122: a temporary of the arithmetic, free while loads are issued; the offset register is
// read when a load is issued), then the three gathers
#define FBP_GOFF16(c) "v_add_u32_e32 v122, %[row], " c "\n\t" "v_lshlrev_b32_e32 v122, 3, v122\n\t"
#define FBP_GOFF32(c) "v_lshlrev_b32_e32 v122, 3, " c "\n\t"
// the same for a gathered vector stored node by node (x, y, z of a node side by side, 24 bytes apart; the three "planes" are then the
// same array at +0, +8, +16 bytes): an irregular mesh's 64 columns of a slot lie in 64 different cache lines, and with planes in 3 x 64
#define FBP_GOFF16X(c) "v_add_u32_e32 v122, %[row], " c "\n\t" "v_mul_u32_u24_e32 v122, 24, v122\n\t"
#define FBP_GOFF32X(c) "v_mul_u32_u24_e32 v122, 24, " c "\n\t"
#define FBP_GATHER(g0, g1, g2)                                             \
  "global_load_dwordx2 " g0 ", v122, %[spl0]\n\t"                          \
  "global_load_dwordx2 " g1 ", v122, %[spl1]\n\t"                          \
  "global_load_dwordx2 " g2 ", v122, %[spl2]\n\t"
#define FBP_GATHER_A FBP_GATHER("v[124:125]", "v[126:127]", "v[128:129]")
#define FBP_GATHER_B FBP_GATHER("v[130:131]", "v[132:133]", "v[134:135]")
// node-by-node vector: x and y of the column in one 16-byte load, z in a second one (two loads per slot instead of three)
#define FBP_GATHERX(g01, g2)                                               \
  "global_load_dwordx4 " g01 ", v122, %[spl0]\n\t"                         \
  "global_load_dwordx2 " g2 ", v122, %[spl0] offset:16\n\t"
#define FBP_GATHERX_A FBP_GATHERX("v[124:127]", "v[128:129]")
#define FBP_GATHERX_B FBP_GATHERX("v[130:133]", "v[134:135]")
#define FBP_ROW(y, a, b, c, x0, x1, x2)                                    \
  "v_cvt_f64_f32_e32 v[120:121], " a "\n\t"                                \
  "v_cvt_f64_f32_e32 v[122:123], " b "\n\t"                                \
  "v_mul_f64 v[120:121], " x0 ", v[120:121]\n\t"                           \
  "v_mul_f64 v[122:123], " x1 ", v[122:123]\n\t"                           \
  "v_add_f64 v[120:121], v[120:121], v[122:123]\n\t"                       \
  "v_cvt_f64_f32_e32 v[122:123], " c "\n\t"                                \
  "v_mul_f64 v[122:123], " x2 ", v[122:123]\n\t"                           \
  "v_add_f64 v[120:121], v[120:121], v[122:123]\n\t"                       \
  "v_add_f64 " y ", " y ", v[120:121]\n\t"
#define FBP_COMPUTE_A                                                                          \
  FBP_ROW("%[y0]", "v136", "v137", "v138", "v[124:125]", "v[126:127]", "v[128:129]")           \
  FBP_ROW("%[y1]", "v139", "v140", "v141", "v[124:125]", "v[126:127]", "v[128:129]")           \
  FBP_ROW("%[y2]", "v142", "v143", "v144", "v[124:125]", "v[126:127]", "v[128:129]")
#define FBP_COMPUTE_B                                                                          \
  FBP_ROW("%[y0]", "v145", "v146", "v147", "v[130:131]", "v[132:133]", "v[134:135]")           \
  FBP_ROW("%[y1]", "v148", "v149", "v150", "v[130:131]", "v[132:133]", "v[134:135]")           \
  FBP_ROW("%[y2]", "v151", "v152", "v153", "v[130:131]", "v[132:133]", "v[134:135]")
// One slot with two or more to follow / with one to follow / the last one; X = the set computed from, Y = the other set.
// cX holds C(k) (consumed), cY holds C(k+1).
// (W1..W4: the four counts above for three gathers per slot, "22" "13" "21" "12"; one less each with two)
#define FBP_FULL(LOADC, GOFF, cX, cY, LOADV_Y, GATHER_Y, COMPUTE_X, W1, W2) \
  LOADC(cX) LOADV_Y "s_waitcnt vmcnt(" W1 ")\n\t" GOFF(cY) GATHER_Y "s_waitcnt vmcnt(" W2 ")\n\t" COMPUTE_X
#define FBP_PENULT(GOFF, cY, LOADV_Y, GATHER_Y, COMPUTE_X, W3, W4)         \
  LOADV_Y "s_waitcnt vmcnt(" W3 ")\n\t" GOFF(cY) GATHER_Y "s_waitcnt vmcnt(" W4 ")\n\t" COMPUTE_X
#define FBP_LAST(COMPUTE_X) "s_waitcnt vmcnt(0)\n\t" COMPUTE_X
#define FBP_BODY(LOADC, GOFF) FBP_BODYG(LOADC, GOFF, FBP_GATHER_A, FBP_GATHER_B, "22", "13", "21", "12")
#define FBP_BODYG(LOADC, GOFF, GATHER_A, GATHER_B, W1, W2, W3, W4)                                                                                 \
  /* the scalar operands may have been written by a vector instruction (v_readlane / v_readfirstlane) just before: a vector   \
     memory instruction reading such a register needs 5 wait states, and the compiler does not look into this text */         \
  "s_nop 4\n\t"                                                                                                 \
  /* prologue: C(0) -> cA (v154), C(1) -> cB (v155) if it exists, V(0) -> A, G(0) -> A */                       \
  LOADC("v154")                                                                                                 \
  "s_cmp_lt_i32 %[n], 2\n\t"                                                                                    \
  "s_cbranch_scc1 .Lfbp_one_%=\n\t"                                                                             \
  LOADC("v155")                                                                                                 \
  FBP_LOADV_A                                                                                                   \
  "s_waitcnt vmcnt(10)\n\t"                                                                                     \
  "s_branch .Lfbp_g0_%=\n"                                                                                      \
  ".Lfbp_one_%=:\n\t"                                                                                           \
  FBP_LOADV_A                                                                                                   \
  "s_waitcnt vmcnt(9)\n"                                                                                        \
  ".Lfbp_g0_%=:\n\t"                                                                                            \
  GOFF("v154") GATHER_A                                                                                     \
  /* even slot: compute from A */                                                                               \
  ".Lfbp_even_%=:\n\t"                                                                                          \
  "s_cmp_lt_i32 %[n], 3\n\t"                                                                                    \
  "s_cbranch_scc1 .Lfbp_even_tail_%=\n\t"                                                                       \
  FBP_FULL(LOADC, GOFF, "v154", "v155", FBP_LOADV_B, GATHER_B, FBP_COMPUTE_A, W1, W2)                               \
  "s_sub_i32 %[n], %[n], 1\n\t"                                                                                 \
  /* odd slot: compute from B */                                                                                \
  "s_cmp_lt_i32 %[n], 3\n\t"                                                                                    \
  "s_cbranch_scc1 .Lfbp_odd_tail_%=\n\t"                                                                        \
  FBP_FULL(LOADC, GOFF, "v155", "v154", FBP_LOADV_A, GATHER_A, FBP_COMPUTE_B, W1, W2)                               \
  "s_sub_i32 %[n], %[n], 1\n\t"                                                                                 \
  "s_branch .Lfbp_even_%=\n"                                                                                    \
  ".Lfbp_even_tail_%=:\n\t"                                                                                     \
  "s_cmp_lt_i32 %[n], 2\n\t"                                                                                    \
  "s_cbranch_scc1 .Lfbp_even_last_%=\n\t"                                                                       \
  FBP_PENULT(GOFF, "v155", FBP_LOADV_B, GATHER_B, FBP_COMPUTE_A, W3, W4)                                            \
  FBP_LAST(FBP_COMPUTE_B)                                                                                       \
  "s_branch .Lfbp_end_%=\n"                                                                                     \
  ".Lfbp_even_last_%=:\n\t"                                                                                     \
  FBP_LAST(FBP_COMPUTE_A)                                                                                       \
  "s_branch .Lfbp_end_%=\n"                                                                                     \
  ".Lfbp_odd_tail_%=:\n\t"                                                                                      \
  "s_cmp_lt_i32 %[n], 2\n\t"                                                                                    \
  "s_cbranch_scc1 .Lfbp_odd_last_%=\n\t"                                                                        \
  FBP_PENULT(GOFF, "v154", FBP_LOADV_A, GATHER_A, FBP_COMPUTE_B, W3, W4)                                            \
  FBP_LAST(FBP_COMPUTE_A)                                                                                       \
  "s_branch .Lfbp_end_%=\n"                                                                                     \
  ".Lfbp_odd_last_%=:\n\t"                                                                                      \
  FBP_LAST(FBP_COMPUTE_B)                                                                                       \
  ".Lfbp_end_%=:\n\t"
#define FBP_CLOBBERS                                                                                                                   \
  "memory", "scc", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133",      \
  "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149",      \
  "v150", "v151", "v152", "v153", "v154", "v155"
// clang-format on

// n >= 1 slots (wave-uniform); voff = byte offset of the lane's first value of the first slot from `vals` (advances 2304 per
// slot), coff = byte offset of the lane's column word of the first slot from `cols`; pl0..2 = the three planes of the gathered
// vector; row = the lane's row (16-bit words count from it).  All lanes of the wavefront must be active.
// Pulls the values of the first n (<= 4) streamed slots of the lane's slice into the XCD's L2 and waits for them: 9 x n loads into
// the temporaries, never read.  Called by the wavefronts that idle while wavefront 0 waits for the neighbours' flags -- the memory
// system idles with them -- so that the product finds these lines in L2 and takes that much less from the Infinity Cache.
__device__ __forceinline__ void pipe_prefetch_values(int n, unsigned int voff, const float* vals);

// a pointer every lane holds alike, forced into a scalar register pair (an "s" operand of a value the compiler takes for
// divergent would otherwise be handed over in vector registers)
template <typename T>
__device__ __forceinline__ T* scalar_ptr(T* p) {
  const unsigned long long b = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void pipe_prefetch_values(int n, unsigned int voff, const float* vals) {
  vals = scalar_ptr(vals);
  n = __builtin_amdgcn_readfirstlane(n);
  asm volatile("s_nop 4\n\t"
               "s_cmp_lt_i32 %[n], 1\n\t"
               "s_cbranch_scc1 .Lfbp_pf_end_%=\n\t"
               FBP_LOADV("v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128")
               "s_cmp_lt_i32 %[n], 2\n\t"
               "s_cbranch_scc1 .Lfbp_pf_wait_%=\n\t"
               FBP_LOADV("v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137")
               "s_cmp_lt_i32 %[n], 3\n\t"
               "s_cbranch_scc1 .Lfbp_pf_wait_%=\n\t"
               FBP_LOADV("v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146")
               "s_cmp_lt_i32 %[n], 4\n\t"
               "s_cbranch_scc1 .Lfbp_pf_wait_%=\n\t"
               FBP_LOADV("v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155")
               ".Lfbp_pf_wait_%=:\n\t"
               "s_waitcnt vmcnt(0)\n"
               ".Lfbp_pf_end_%=:\n\t"
               : [voff] "+v"(voff)
               : [n] "s"(n), [svals] "s"(vals)
               : FBP_CLOBBERS);
}

// XYZ: the gathered vector is stored node by node (pl1 = pl0 + 1, pl2 = pl0 + 2 doubles; columns below 2^24) instead of in three planes
template <bool C16, bool XYZ = false>
__device__ __forceinline__ void pipe_stream_slots(int n, unsigned int voff, unsigned int coff, const float* vals, const void* cols, const double* pl0,
                                                  const double* pl1, const double* pl2, int row, double& y0, double& y1, double& y2) {
  vals = scalar_ptr(vals); cols = scalar_ptr(cols); pl0 = scalar_ptr(pl0); pl1 = scalar_ptr(pl1); pl2 = scalar_ptr(pl2);
  n = __builtin_amdgcn_readfirstlane(n);
  if (C16 && XYZ) {
    asm volatile(FBP_BODYG(FBP_LOADC16, FBP_GOFF16X, FBP_GATHERX_A, FBP_GATHERX_B, "21", "12", "20", "11")
                 : [y0] "+v"(y0), [y1] "+v"(y1), [y2] "+v"(y2), [voff] "+v"(voff), [coff] "+v"(coff), [n] "+s"(n)
                 : [svals] "s"(vals), [scols] "s"(cols), [spl0] "s"(pl0), [spl1] "s"(pl1), [spl2] "s"(pl2), [row] "v"(row)
                 : FBP_CLOBBERS);
  } else if (XYZ) {
    asm volatile(FBP_BODYG(FBP_LOADC32, FBP_GOFF32X, FBP_GATHERX_A, FBP_GATHERX_B, "21", "12", "20", "11")
                 : [y0] "+v"(y0), [y1] "+v"(y1), [y2] "+v"(y2), [voff] "+v"(voff), [coff] "+v"(coff), [n] "+s"(n)
                 : [svals] "s"(vals), [scols] "s"(cols), [spl0] "s"(pl0), [spl1] "s"(pl1), [spl2] "s"(pl2), [row] "v"(row)
                 : FBP_CLOBBERS);
  } else if (C16) {
    asm volatile(FBP_BODY(FBP_LOADC16, FBP_GOFF16)
                 : [y0] "+v"(y0), [y1] "+v"(y1), [y2] "+v"(y2), [voff] "+v"(voff), [coff] "+v"(coff), [n] "+s"(n)
                 : [svals] "s"(vals), [scols] "s"(cols), [spl0] "s"(pl0), [spl1] "s"(pl1), [spl2] "s"(pl2), [row] "v"(row)
                 : FBP_CLOBBERS);
  } else {
    asm volatile(FBP_BODY(FBP_LOADC32, FBP_GOFF32)
                 : [y0] "+v"(y0), [y1] "+v"(y1), [y2] "+v"(y2), [voff] "+v"(voff), [coff] "+v"(coff), [n] "+s"(n)
                 : [svals] "s"(vals), [scols] "s"(cols), [spl0] "s"(pl0), [spl1] "s"(pl1), [spl2] "s"(pl2), [row] "v"(row)
                 : FBP_CLOBBERS);
  }
}

}  // namespace fb
