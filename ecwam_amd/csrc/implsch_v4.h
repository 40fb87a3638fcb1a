// IMPLSCH (implsch.F90:10-468 and the routines it inlines), the product's one kernel generation since round 5: PP sea points per wavefront,
// G = NANG/2 lanes per point, lane j of a point owns the ADJACENT direction pair (K = 2j, 2j+1).  Single and double precision,
// NANG = 48 / 36 / 24 / 12 (PP = 2 / 3 / 5 / 10 in single precision), NFRE = 36.
//
// What changed against the round-1 three-points-per-wavefront kernel (k_implsch3: pairs (K, K+18), ds_bpermute rotations), and why (tools/ubench_valu.hip, profiles/r02_ubench_valu.txt, MI355X):
//   * ds_bpermute_b32 costs 6 cycles of the CU's single LDS pipe per wave instruction, ds_read_b32 / ds_read_b64 cost 2: the 53
//     bpermutes per interaction frequency of k_implsch3 kept the LDS pipe ~70 % busy during the sweep.  Here every rotation in K
//     is a ds_read_b64 of an LDS row at a per-lane wrapped address (the lane's pair shifted by an even number of directions;
//     odd shifts take one half from each of two even shifts): the DIA gathers read four staged rows (the frequency-interpolated
//     spectra), the DIA increments are staged in three rows and pulled back rotated, the 2 NSDSNTH + 1 taps of the saturation
//     filter are NSDSNTH + 1 reads of the row itself.  No half swaps (v_cndmask), no per-tap address registers.
//   * the arithmetic is written on pairs (ext_vector_type(2): the two directions of a lane, or the two gust states of a row) and
//     compiles to v_pk_* in single precision; v_pk_fma_f32 issues at 4.4 cycles per wave instruction against 2.6 for v_fma_f32 when
//     two waves share a SIMD, so packing pays through the instruction count (the same source is plain arithmetic in double precision).
//   * the tile is [M][point][K] (K fastest, natural order): the coalesced load / store are 16-byte global accesses with the
//     (K, M) transposition done by four 4-byte LDS accesses at immediate offsets -- 5 instead of 45 vector instructions per element.
//   * the planes SQRT(WAVNUM) and LOG(WAVNUM Z0M) of the factor table double as staging rows during the sweep, the staging rows hold
//     SINPUT's row integrals outside it: 20 424 B of LDS per wave (sp, NANG = 36), 8 waves per CU as before -- the limit: 24 sea
//     points per CU is what 160 KB hold.
//   * with two waves per SIMD a wave issues one instruction per ~5.3 cycles whatever its kind, and a third of its life was spent in
//     s_waitcnt: the sweep is software-pipelined (three LDS round trips per interaction frequency instead of eight, see there), no
//     store is predicated, and the row loops are straight-line code: a uniform branch per row costs more than most of the work such
//     branches skip.  Constants per frequency come as ONE scalar-loaded record per row (DevTab::SINROW, DIACF).
//   * the dependent scalar chains that need no spectrum on either side run one sea point per lane in k_implsch4_pre (first TAUT_Z0)
//     and k_implsch4_fin (second STRESSO / TAU_PHI_HF, WNFLUXES, NEMO coupling outputs) around the main kernel.
// The lane-per-point scalar stages between the vector stages (STRESSO of the first call, second TAUT_Z0, WSIGSTAR, swell set-up,
// SDIWBK) are those of implsch_point.h.
#pragma once
#include "ctu.h"

#define V4_NFRE 36
// phase timing (tools/time_v4_phases.sh): a diagnostics build (-DECWAM_HIP_DIAGNOSTICS) returns early at the phase boundary DBG_SKIP
#ifdef ECWAM_HIP_DIAGNOSTICS
#define V4_PHASE_EXIT(k) do { if (tb.DBG_SKIP == (k)) return; } while (0)
#else
#define V4_PHASE_EXIT(k) do { } while (0)
#endif
// index checks of the debugging build (-DV4_CHECK=1, build variant "rdpchk"): device-side assert = message + trap
#if defined(V4_CHECK) && V4_CHECK
#include <cassert>
#define V4_CHK(c) assert(c)
#else
#define V4_CHK(c) do { } while (0)
#endif
#define V4_NSTG 4
// frequencies per lane: lane j of a point owns M = q G + j + 1, q = 0 .. NS - 1 (36 directions: 2, 24: 3, 12: 6; 48 directions: 2 with the
// second one only on the lanes j < NFRE - G = 12 -- the other lanes repeat frequency NFRE: the same stores, a weight of zero in the sums)
#define V4_NS(nang) ((V4_NFRE + (nang) / 2 - 1) / ((nang) / 2))
#define V4_PLN(pp, nang) ((pp) * ((nang) > V4_NFRE ? (nang) : V4_NFRE))
#define V4_NFAC 6   // words per (point, frequency) of the factor table: [M][4] BSC, SBO, CINV, WAVNUM + the planes SQ and ZCN

#define V4SYNC() WSYNC()
#define V4_WPE_MIN(T) (sizeof(T) == 4 ? 2 : 1)

template <typename T>
using V2 = T __attribute__((ext_vector_type(2)));

// the row of scalars per sea point that k_implsch4_pre hands to k_implsch4 and k_implsch4 to k_implsch4_fin
enum { FIN_AIRD = 0, FIN_UFRIC, FIN_Z0M, FIN_MIJ, FIN_XS, FIN_YS, FIN_F1DCOS3, FIN_F1DCOS2, FIN_F1DSIN2, FIN_F1D, FIN_RNFAC, FIN_PHIWA,
       FIN_SINWD, FIN_COSWD, FIN_WSWAVE, FIN_CICOVER, FIN_PHILF, FIN_XSTRESS, FIN_YSTRESS, FIN_Z0B, FIN_CHRNCK, FIN_COSDIFF, FIN_EMEAN, FIN_F1MEAN,
       FIN_TAUICX, FIN_TAUICY, FIN_STRNMS, FIN_SPARE,                   // (these four: the RARE build's ice stress and strain)
       // the two-kernel split (PART = 1 -> PART = 2): means of the spectrum before the update, the windsea mean frequency of the second
       // SINFLX call, SDIWBK's rate and the scale of SDEPTHLIM (the second kernel re-applies it to the spectrum it loads)
       FIN_FMEAN, FIN_FMEANWS, FIN_AKMEAN, FIN_XKMEAN, FIN_SDS, FIN_SC, V4_NFIN = 36 };

__device__ __forceinline__ float v4_bp(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ double v4_bp(int addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ V2<T> v4_same(V2<T> v, int addr) {
  V2<T> r;
  r.x = v4_bp(addr, v.x);
  r.y = v4_bp(addr, v.y);
  return r;
}
// all-reduce over the G lanes of a point.
//   G = 18 (36 directions, three points per wave): lanes 0..47 hold pairs 0..15 of the three points, one point per DPP row of 16
//   lanes; lanes 48..53 hold pairs 16, 17 ("extras", two lanes per point).  A reduction folds the extras into lanes 0, 1 of the row
//   (one ds_bpermute: a0 = the extra of lanes 0, 1, the lane itself elsewhere, fold = 1 / 0), rotates inside the row (v_*_dpp
//   row_ror 8, 4, 2, 1: no LDS) and hands the total to the extras and the shadows (one ds_bpermute: a1 = lane 0 of the row for
//   them, the lane itself in rows 0..2).  ds_bpermute costs 6 cycles of the CU's LDS pipe: 2 per quantity instead of 5.
//   G = 24 / 12 / 6: rotations by 12, 6, 3, then 1 and 2 / by 6, 3, then 1 and 2 / by 3, then 1 and 2 (byte addresses of the source lanes).
template <typename T>
struct V4Rot { int a0, a1, a2, a3, a4; T fold; };
template <int CTRL>
__device__ __forceinline__ float v4_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double v4_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
#define V4_ROW_ROR(n) (0x120 + (n))
template <typename T>
__device__ __forceinline__ V2<T> v4_rowsum(V2<T> v) {
  T x = v.x, y = v.y;   // component-wise: v_add_f32_dpp takes the rotated operand directly, a packed add would need two v_mov_dpp first
  x += v4_dpp<V4_ROW_ROR(8)>(x); y += v4_dpp<V4_ROW_ROR(8)>(y);
  x += v4_dpp<V4_ROW_ROR(4)>(x); y += v4_dpp<V4_ROW_ROR(4)>(y);
  x += v4_dpp<V4_ROW_ROR(2)>(x); y += v4_dpp<V4_ROW_ROR(2)>(y);
  x += v4_dpp<V4_ROW_ROR(1)>(x); y += v4_dpp<V4_ROW_ROR(1)>(y);
  return V2<T>{x, y};
}
template <typename T>
__device__ __forceinline__ T v4_rowmax(T v) {
  v = m_max(v, v4_dpp<V4_ROW_ROR(8)>(v));
  v = m_max(v, v4_dpp<V4_ROW_ROR(4)>(v));
  v = m_max(v, v4_dpp<V4_ROW_ROR(2)>(v));
  v = m_max(v, v4_dpp<V4_ROW_ROR(1)>(v));
  return v;
}
// single precision: v_max_f32_dpp directly (the compiler keeps v_mov_dpp + a canonicalising v_max + v_max per step); the s_nop covers
// the two wait states between a VALU write of a register and its use as a DPP operand
template <>
__device__ __forceinline__ float v4_rowmax<float>(float v) {
  asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "+v"(v));
  return v;
}
template <int G, typename T>
__device__ __forceinline__ V2<T> v4_allsum(V2<T> v, const V4Rot<T>& r) {
  if constexpr (G == 18) {
    v = v + r.fold * v4_same<T>(v, r.a0);
    v = v4_rowsum<T>(v);
    return v4_same<T>(v, r.a1);
  } else {
    v = v + v4_same<T>(v, r.a0);
    if (G >= 12) v = v + v4_same<T>(v, r.a1);
    if (G == 24) v = v + v4_same<T>(v, r.a2);
    v = v + (v4_same<T>(v, r.a3) + v4_same<T>(v, r.a4));
    return v;
  }
}
// one quantity (the other half of a pair would be wasted exchanges)
template <int G, typename T>
__device__ __forceinline__ T v4_allsum1(T v, const V4Rot<T>& r) {
  if constexpr (G == 18) {
    v = v + r.fold * v4_bp(r.a0, v);
    v = v + v4_dpp<V4_ROW_ROR(8)>(v);
    v = v + v4_dpp<V4_ROW_ROR(4)>(v);
    v = v + v4_dpp<V4_ROW_ROR(2)>(v);
    v = v + v4_dpp<V4_ROW_ROR(1)>(v);
    return v4_bp(r.a1, v);
  } else {
    v = v + v4_bp(r.a0, v);
    if (G >= 12) v = v + v4_bp(r.a1, v);
    if (G == 24) v = v + v4_bp(r.a2, v);
    v = v + (v4_bp(r.a3, v) + v4_bp(r.a4, v));
    return v;
  }
}

// N independent all-reduces at once (V4_REDN): the exchanges of every stage are issued together and waited for together -- N quantities
// cost the LDS round trips of one.  The first NB come back on every lane of the point; the others (G = 18) are complete on the sixteen
// lanes of the point's DPP row only, which is what v4_row_total_to_lds needs.  Per quantity the operations and their order are those of
// v4_allsum / v4_allsum1 / v4_row_total_to_lds: the same bits.
template <int G, int N, int NB, typename T>
__device__ __forceinline__ void v4_allsum_n(T (&v)[N], const V4Rot<T>& r) {
  T e[N];
  if constexpr (G == 18) {
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = v4_bp(r.a0, v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + r.fold * e[i];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + v4_dpp<V4_ROW_ROR(8)>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + v4_dpp<V4_ROW_ROR(4)>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + v4_dpp<V4_ROW_ROR(2)>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + v4_dpp<V4_ROW_ROR(1)>(v[i]);
#pragma unroll
    for (int i = 0; i < NB; i++) e[i] = v4_bp(r.a1, v[i]);
#pragma unroll
    for (int i = 0; i < NB; i++) v[i] = e[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = v4_bp(r.a0, v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + e[i];
    if constexpr (G >= 12) {
#pragma unroll
      for (int i = 0; i < N; i++) e[i] = v4_bp(r.a1, v[i]);
#pragma unroll
      for (int i = 0; i < N; i++) v[i] = v[i] + e[i];
    }
    if constexpr (G == 24) {
#pragma unroll
      for (int i = 0; i < N; i++) e[i] = v4_bp(r.a2, v[i]);
#pragma unroll
      for (int i = 0; i < N; i++) v[i] = v[i] + e[i];
    }
    T f[N];
#pragma unroll
    for (int i = 0; i < N; i++) { e[i] = v4_bp(r.a3, v[i]); f[i] = v4_bp(r.a4, v[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = v[i] + (e[i] + f[i]);
  }
}
#ifndef V4_REDN
#define V4_REDN 1
#endif

// maximum over the lanes of a point (the RARE build's PEAK_ANG)
template <int G, typename T>
__device__ __forceinline__ T v4_allmax1(T v, const V4Rot<T>& r) {
  if constexpr (G == 18) {
    v = m_max(v, v4_bp(r.a0, v));      // lanes 0, 1 of a row take their point's extras in, the others themselves
    v = v4_rowmax<T>(v);
    return v4_bp(r.a1, v);
  } else {
    v = m_max(v, v4_bp(r.a0, v));
    if (G >= 12) v = m_max(v, v4_bp(r.a1, v));
    if (G == 24) v = m_max(v, v4_bp(r.a2, v));
    return m_max(v, m_max(v4_bp(r.a3, v), v4_bp(r.a4, v)));
  }
}

// The positive wind input of a row summed over the directions of the point -> the point's row table in LDS (SINPUT's second call;
// weighted below the cut-off once MIJ is known, stresso.F90:160-168).  36 directions: the extras are folded into lanes 0, 1 of the point's
// DPP row, four row rotations, every lane of the row holds the total and stores it (the same value at the same address); the extras'
// own row of 16 lanes sums a mixture nobody reads: their store goes to a private slot (spw).  No trip through global memory (rounds 1
// and 2 parked the per-lane shares in the point's FL1 block: 1.1 GB written and read back per O320 launch).
template <int G, typename T>
__device__ __forceinline__ void v4_row_total_to_lds(T v, const V4Rot<T>& r, T* spw, int m) {
  if constexpr (G == 18) {
    v = v + r.fold * v4_bp(r.a0, v);
    v = v + v4_dpp<V4_ROW_ROR(8)>(v);
    v = v + v4_dpp<V4_ROW_ROR(4)>(v);
    v = v + v4_dpp<V4_ROW_ROR(2)>(v);
    v = v + v4_dpp<V4_ROW_ROR(1)>(v);
  } else {
    v = v4_allsum1<G, T>(v, r);
  }
  spw[m] = v;
}

// the pair (X(2j+r), X(2j+r+1)) of an LDS row; sh[i] = index of element (2j + 2(i-NSH)) mod NANG of the lane's point in a row
template <typename T, int NSH, int r>
__device__ __forceinline__ V2<T> v4_at(const T* row, const int (&sh)[2 * NSH + 1]) {
  if constexpr ((r & 1) == 0) {
    return *reinterpret_cast<const V2<T>*>(row + sh[r / 2 + NSH]);
  } else {
    // two 4-byte reads straight into the halves of the pair (two 8-byte reads need a v_pk_mov_b32 to assemble it: 4.4 issue cycles,
    // the cost of a packed multiply-add)
    return V2<T>{row[sh[(r - 1) / 2 + NSH] + 1], row[sh[(r + 1) / 2 + NSH]]};
  }
}

// ca X(ra) + cb X(rb) for two ADJACENT rotations (|ra - rb| = 1) of an LDS row, X(r) = the pair (X(2j+r), X(2j+r+1)): the three elements
// involved lie in the two ALIGNED pairs at the even shifts e and e + 2, e = 2 floor(min(ra, rb) / 2) -- two ds_read_b64 (one, where a shift is
// zero: `own` is the lane's pair of that row, already in registers) and four plain multiply-adds on the halves.  Round 5: the DIA's angular
// interpolation weights are the same for every interaction (inisnonlin.F90:186-241), so a gather / scatter pair of an interaction is ONE
// such window of ONE staged row instead of an odd rotation (two 4-byte reads, two-way bank conflicts by construction) plus an even one of
// two rows: 28 % fewer LDS instructions in the sweep, which profiles/r05_bench_O320_sp_split_pmc.json shows to be bound by the LDS
// array (88 % busy, vector ALU 55 %).
// V4_RECPF: the per-interaction coefficient record of the sweep (DevTab::DIAREC, 19 words) is fetched ONE INTERACTION AHEAD by three scalar
// loads at the top of an interaction and pinned in scalar registers at its end.  Left to itself the compiler sinks each scalar load to the
// block of its first use: eight loads in six places per interaction, each with its s_waitcnt lgkmcnt(0) one or two instructions later -- the
// scalar-cache latency exposed six times per interaction, and the LDS reads in flight drained with it (scalar loads return out of order).
// The 19 + 19 registers come from moving the sweep's loop-invariant uniform weights (saturation filter, angular weights) to vector registers.
#ifndef V4_RECPF
#define V4_RECPF 1
#endif
// Double precision (V4_RECV = 2): two records are 76 scalar registers, so the record of THIS interaction is fetched by scalar loads issued
// together at its top and held there by a scheduling barrier: five loads in one place and 70 lgkmcnt(0) waits per eight interactions instead
// of seven in six places and 91; - 4 % kernel time, bit-identical (0: as the compiler places them; the record one interaction ahead in vector
// registers was measured in round 5: + 28 % vector instructions, the same time -- profiles/r05_scalar_prefetch.txt).
#ifndef V4_RECV
#define V4_RECV 2
#endif
// Single precision only: in double precision the two records are 80 scalar registers, and the variant measured 3 % slower there.
// The builds with more live registers (48 directions: twelve filter weights; the RARE builds) spill to scratch with it and stay without.
template <bool ON, typename T>
__device__ __forceinline__ T v4_vreg(T x) {      // a wave-uniform value moved to a vector register on purpose
  if constexpr (ON && sizeof(T) == 4) {
    T r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
  } else return x;
}
template <typename T>
__device__ __forceinline__ void v4_pin6(const T (&r)[6]) { asm volatile("" :: "s"(r[0]), "s"(r[1]), "s"(r[2]), "s"(r[3]), "s"(r[4]), "s"(r[5])); }
template <typename T>
__device__ __forceinline__ void v4_pin10(T (&r)[20], int o) {      // ten values held in scalar registers at this point of the program
  // (inputs only: the values must EXIST in scalar registers here, before their uses in the next interaction -- their loads cannot sink past this
  // point; as outputs they would be copied out of the loads' register tuples one by one)
  asm volatile("" :: "s"(r[o]), "s"(r[o + 1]), "s"(r[o + 2]), "s"(r[o + 3]), "s"(r[o + 4]), "s"(r[o + 5]), "s"(r[o + 6]), "s"(r[o + 7]), "s"(r[o + 8]), "s"(r[o + 9]));
}
// A STRADDLING pair (X(2j+r), X(2j+r+1)), r odd -- the high half of one aligned pair and the low half of the next (the DIA windows, 12 per
// interaction, and the odd taps of the saturation filter) -- is left to the vectoriser (two v_mov_b32 each).  One shuffle per pair, or no pair
// assembled at all (the consuming operation as two plain operations on the halves), were measured in round 5: the same time within 1 %, not
// bit-identical (profiles/r05_pair_shuffle_ab.txt).
template <typename T, int NSH, int ra, int rb>
__device__ __forceinline__ V2<T> v4_win(const T* row, const int (&sh)[2 * NSH + 1], V2<T> own, T ca, T cb) {
  static_assert(ra - rb == 1 || rb - ra == 1, "adjacent rotations");
  constexpr int lo = ra < rb ? ra : rb;
  constexpr int e = (lo >= 0) ? (lo / 2) * 2 : -(((-lo) + 1) / 2) * 2;      // 2 floor(lo / 2)
  static_assert(e / 2 + NSH >= 0 && e / 2 + 1 + NSH <= 2 * NSH, "shift table");
  const V2<T> a = (e == 0) ? own : *reinterpret_cast<const V2<T>*>(row + sh[e / 2 + NSH]);
  const V2<T> b = (e + 2 == 0) ? own : *reinterpret_cast<const V2<T>*>(row + sh[e / 2 + 1 + NSH]);
  const T el[4] = {a.x, a.y, b.x, b.y};      // elements e .. e + 3
  return V2<T>{ca * el[ra - e] + cb * el[rb - e], ca * el[ra - e + 1] + cb * el[rb - e + 1]};
}

// the pair (X(2j+r), X(2j+r+1)) out of the aligned pairs A[i] = (X(2j + 2(i-NSH)), X(2j + 2(i-NSH) + 1)) of a row already in registers: an aligned
// pair, or one shuffle of two neighbours
template <typename T, int NSH, int r>
__device__ __forceinline__ V2<T> v4_pair(const V2<T> (&A)[2 * NSH + 1]) {
  static_assert(r >= -2 * NSH && r <= 2 * NSH, "window");
  if constexpr ((r & 1) == 0) return A[NSH + r / 2];
  else return __builtin_shufflevector(A[NSH + (r - 1) / 2], A[NSH + (r + 1) / 2], 1, 2);
}
// taps d .. NH of the symmetric saturation filter: acc += wt[NH - d] (X(-d) + X(d)) on pairs (sdissip_ard.F90:142-174)
template <typename T, int NSH, int NH, int d>
__device__ __forceinline__ void v4_sat_taps(const V2<T> (&A)[2 * NSH + 1], const T (&wt)[NH + 1], V2<T>& acc) {
  if constexpr (d <= NH) {
    acc += wt[NH - d] * (v4_pair<T, NSH, -d>(A) + v4_pair<T, NSH, d>(A));
    v4_sat_taps<T, NSH, NH, d + 1>(A, wt, acc);
  }
}

template <typename T, int PP, bool RARE>
__device__ __forceinline__ void v4_stresso(const DevTab<T>& tb, T* sSC, int lane, bool LLPHIWA) {
  if constexpr (PP <= 3) {
    stresso_stage<T, PP, RARE>(tb, sSC, lane, LLPHIWA);
  } else {
    stresso_stage<T, 3, RARE>(tb, sSC, lane, LLPHIWA);
    WSYNC();
    v4_stresso<T, PP - 3, RARE>(tb, sSC + 3 * NSC, lane, LLPHIWA);
  }
}

template <typename T, int NANG_, int PP_>
struct V4Ctx {
  int lane, p, j;
  T* tile;        // row 0 of the wave's tile [M][point][K]
  int own;        // p NANG + 2j: the lane's pair inside a row
  T* fac4;        // [M][4] of the point: BSC, SBO, CINV, WAVNUM
  T* sq;          // [M] SQRT(WAVNUM)
  T* zcn;         // [M] LOG(WAVNUM Z0M) of the current SINFLX call
  T* c;           // scalars of the point [NSC]
  V4Rot<T> rot;
  V2<T> sinth, costh;
  // module tables per frequency, lane m holds M = m+1: broadcast with v_readlane inside the M loops
  T rDFIM, rDFIMOFR, rDFIMFR, rZPIFR, rRHOWG, rCOFRM4, rFLMAX;
};
enum { Q4_BSC = 0, Q4_SBO, Q4_CINV, Q4_WAVNUM };   // Q4_BSC = WAVNUM XK2CG / 2 pi

// ---- The advecting tile load (round 6, ADV builds of k_implsch4): PROPAGS2 (propags2.F90:99-121, IREFRA = 0) of the wave's PP points from the
//      rows of A.f_in straight into the tile [M][point][K], so that a 1:1 WAMINTGR step (wamintgr.F90:94-146: PROPAG_WAM, NEWWIND, IMPLSCH) is
//      ONE pass over the spectra instead of two -- the advected spectrum never goes to memory.  The arithmetic is ctu.h's, the helpers
//      k_propags2_otf applies (contraction off): the tile holds bit for bit what that kernel would have written to FL3.
//      Work order: a "chunk" is the 16 bytes (direction k, frequencies m .. m + VEC - 1) of one point, chunk w = lane + 64 it of point q is
//      one step; a step's eight 16-byte gathers (own, longitude, two latitude, two corner neighbours, directions k -+ 1 of the own row) are
//      issued D steps ahead of the step that consumes them (a ring of D x 8 vectors in registers), so that the memory latency of a step
//      hides behind the weights and stencil of the D - 1 steps before it.  Per-point scalars of the weights come from the table k_ctu_prep
//      (propag.hip) fills -- their divisions are correctly rounded there, this translation unit is compiled with the fast ones.
template <typename T>
struct V4Adv {
  const T* f_in;        // [rows][NANG][NFRE]: owned rows, halo rows, land row -- what the stencil reads (never the rows the kernel stores to)
  const int* klon;      // [n][2]
  const int* klat;      // [n][2][2]
  const int* kcor;      // [n][4][2]
  const T* cg;          // CGROUP_EXT [rows][NFRE]
  const T* pt;          // [n][12]: ZDELLO, COSPHM1, 1 / (ZDELLO XDELLA), TAN(lat), DP(1:2), WLAT(1:2), WCOR(1:4) (k_ctu_prep)
  const T* dirT;        // [NANG][4]: DELTH0 (SINTH(K) + SINTH(K+1)) / R, the same for K-1, SINTH, COSTH; then CMTODEG
  const int* dirI;      // [NANG][4]: JXO(K,1) | JYO(K,1) << 1 | KCR(K,1) << 2, KPM(K,-1), KPM(K,1)
  T xdella, delpro;
  // ADV = 3 (the native O1280 mode, propag_wam.F90:247-313): frequencies [0, mlf) -- the fast waves -- advance with delpro_lf, and the first
  // gin_k frequencies of every direction (own and neighbours) are READ from the compact rows gin[rows][NANG][gin_k], the fast waves after
  // their sub-steps; dirT then carries the factors of ctu_dirfac for delpro_lf behind CMTODEG ([NANG][2])
  const T* gin;
  T delpro_lf;
  int gin_k, mlf;
  // ADV & 4 (LSUBGRID, the reference's default on real bathymetry): OBS[n][8][NFRE], the transmission coefficients OBSLAT(1:2), OBSLON(1:2),
  // OBSCOR(1:4) that scale the space weights of the neighbours (ctuw.F90:703-733)
  const T* obs;
  int m0, m1;           // advected frequencies [m0, m1); the others are carried over
  int xcd_walk;         // XCD-aware order of the workgroups (diagnostics: measured 1 % slower than the natural order, profiles/r06_fused_*.txt)
};
#ifndef V4_ADV_DEPTH
#define V4_ADV_DEPTH 3
#endif
template <typename T, int NANG, int PP, int MODE, int NSCR>
__device__ __forceinline__ void v4_advect_tile(const V4Adv<T>& A, int ij0, int n, int lane, T* __restrict__ sT, T* __restrict__ sScr) {
  constexpr int NFRE = V4_NFRE, N = NANG * NFRE, RS = PP * NANG, VEC = 16 / (int)sizeof(T), NC = NFRE / VEC, NVL = N / VEC;
  // steps: NFULL rounds of 64 chunks per point, point after point inside a round (the chunk's direction and frequencies are then the same
  // for the PP steps of a round); the NREM chunks of every point that are left share NTS tail steps (36 directions, single precision:
  // 324 chunks per point = 5 rounds + 4, the 12 left-over chunks of the three points are ONE step: 16 steps instead of 18)
  constexpr int NFULL = NVL / 64, NREM = NVL - 64 * NFULL, NTS = (PP * NREM + 63) / 64, NCH = NFULL * PP + NTS, D = V4_ADV_DEPTH;
  constexpr int PTW = 20;                              // words per point of sPt
  typedef T VT __attribute__((ext_vector_type(VEC)));
  typedef int I4 __attribute__((ext_vector_type(4)));
  T* sB = sScr;                                        // [PP][5][NFRE]: |h(1:2)|, |hy(1:2)|, CG of ctu_base
  T* sPt = sB + PP * 5 * NFRE;                         // [PP][PTW]: ZDELLO, |COSPHM1|, GA, TANPH, DP(1:2), WLAT(1:2), WCOR(1:4), 1 - WLAT, 1 - WCOR
  T* sK = sPt + PP * PTW;                              // [NANG][4]: 2 SP, 2 SM (ctu_dirfac with TANPH = 1, doubled), |SINTH|, |COSTH|; then CMTODEG
  constexpr bool LF = (MODE & 3) == 3;                 // fast waves with their own time step, read from compact rows
  constexpr bool OBS = (MODE & 4) != 0;                // sub-grid obstructions: three more 16-byte loads per step (the coefficients of the step's neighbours)
  constexpr int NB = OBS ? 11 : 8;
  static_assert(MODE == 2 || (MODE & 1), "modes: 1 plain, 3 fast waves, + 4 obstructions; 2 the probe");
  constexpr int NKW = NANG * 4 + 4 + (LF ? NANG * 2 : 0);      // (LF: then [NANG][2]: 2 SP, 2 SM for DELPRO_LF)
  int* sI = reinterpret_cast<int*>(sK + NKW);          // [PP][16]: ij, KLON(1:2), KLAT(1:2,1:2), KCOR(1:4,1:2)
  int* sD = sI + PP * 16;                              // [NANG][4]
  static_assert((PP * 5 * NFRE + PP * PTW + NKW) * sizeof(T) + (PP * 16 + NANG * 4) * sizeof(int) <= NSCR * sizeof(T), "LDS scratch of the advecting load");
  for (int i = lane; i < PP * PTW; i += 64) {
    const int q = i / PTW, e = i - q * PTW;
    const T* g = A.pt + (size_t)(ij0 + (q < n ? q : n - 1)) * 12;
    T v = T(0);
    if (e < 12) v = g[e];
    else if (e < 18) v = T(1) - g[e - 6];              // 1 - WLAT(1:2), 1 - WCOR(1:4)
    else if (e == 18) v = g[0] * g[2];                 // ZDELLO GA
    else v = A.xdella * g[2];                          // XDELLA GA
    if (e == 1) v = m_abs(v);
    sPt[i] = v;
  }
  for (int i = lane; i < PP * 16; i += 64) {
    const int q = i >> 4, e = i & 15;
    const int ijq = ij0 + (q < n ? q : n - 1);
    int v = ijq;
    if (e >= 1 && e <= 2) v = A.klon[(size_t)ijq * 2 + (e - 1)];
    if (e >= 3 && e <= 6) v = A.klat[(size_t)ijq * 4 + (e - 3)];
    if (e >= 7 && e <= 14) v = A.kcor[(size_t)ijq * 8 + (e - 7)];
    sI[i] = v;
  }
  for (int i = lane; i < NKW; i += 64) {
    const T v = A.dirT[i];
    const int e = i & 3;
    sK[i] = i >= NANG * 4 + 4 ? T(2) * v : (i >= NANG * 4 ? v : (e < 2 ? T(2) * v : m_abs(v)));
  }
  for (int i = lane; i < NANG * 4; i += 64) sD[i] = A.dirI[i];
  WSYNC();
#if defined(V4_ADV_PRIO)
  __builtin_amdgcn_s_setprio(V4_ADV_PRIO);      // experiment: the wave that has gathers to issue goes first on its SIMD
#endif
  VT buf[D][NB];
  // (point, direction, first frequency) of the lane's chunk in step c
  auto chunk_of = [&](int c, int& q, int& k, int& m) {
    int w;
    if (c < NFULL * PP) {
      const int it = c / PP;
      q = c - it * PP;
      w = lane + 64 * it;
    } else {      // tail steps: chunk t of the PP NREM left-over ones; the lanes beyond the last one repeat it (the same stores)
      int t = lane + 64 * (c - NFULL * PP);
      t = t < PP * NREM ? t : PP * NREM - 1;
      q = t / (NREM > 0 ? NREM : 1);
      w = 64 * NFULL + (t - q * NREM);
    }
    k = w / NC;
    m = (w - k * NC) * VEC;
  };
  auto issue = [&](int c, VT (&b)[NB]) {
    int q, k, m;
    chunk_of(c, q, k, m);
    const I4 dk = *reinterpret_cast<const I4*>(sD + 4 * k);
    const int jx0 = dk.x & 1, jy0 = (dk.x >> 1) & 1, kc = (dk.x >> 2) & 3;
    const int* iq = sI + q * 16;
    // (LF: this chunk's operands, own and neighbours, live in the compact rows when its frequencies are among their gin_k)
    const bool fromg = LF && m < A.gin_k;
    const T* src = fromg ? A.gin : A.f_in;
    const int rk = fromg ? A.gin_k : NFRE;               // frequencies per direction of the source rows
    const size_t rn = (size_t)NANG * rk;
    const T* own = src + (size_t)iq[0] * rn;
    const int el = k * rk + m;
    b[0] = *reinterpret_cast<const VT*>(own + el);
    b[1] = *reinterpret_cast<const VT*>(src + (size_t)iq[1 + jx0] * rn + el);
    b[2] = *reinterpret_cast<const VT*>(src + (size_t)iq[3 + 2 * jy0] * rn + el);
    b[3] = *reinterpret_cast<const VT*>(src + (size_t)iq[4 + 2 * jy0] * rn + el);
    b[4] = *reinterpret_cast<const VT*>(src + (size_t)iq[7 + 2 * kc] * rn + el);
    b[5] = *reinterpret_cast<const VT*>(src + (size_t)iq[8 + 2 * kc] * rn + el);
    b[6] = *reinterpret_cast<const VT*>(own + dk.y * rk + m);
    b[7] = *reinterpret_cast<const VT*>(own + dk.z * rk + m);
    if constexpr (OBS) {
      const T* o = A.obs + (size_t)iq[0] * 8 * NFRE + m;
      b[8] = *reinterpret_cast<const VT*>(o + (2 + jx0) * NFRE);      // OBSLON(JXO(K,1))
      b[9] = *reinterpret_cast<const VT*>(o + jy0 * NFRE);            // OBSLAT(JYO(K,1))
      b[10] = *reinterpret_cast<const VT*>(o + (4 + kc) * NFRE);      // OBSCOR(KCR(K,1))
    }
  };
  auto finish = [&](int c, const VT (&b)[NB]) {
    int q, k, m;
    chunk_of(c, q, k, m);
    T r[VEC];
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < VEC; i++) r[i] = T(0.3) * b[0][i] + T(0.1) * (((b[1][i] + b[2][i]) + (b[3][i] + b[4][i])) + ((b[5][i] + b[6][i]) + b[7][i]));
    } else {
      const int sel = sD[4 * k];
      const int jx0 = sel & 1, jy0 = (sel >> 1) & 1, kc = (sel >> 2) & 3;
      const T* pq = sPt + q * PTW;
      const T zd = pq[0], acpm1 = pq[1], ga = pq[2], tanph = pq[3], wl = pq[6 + jy0], omwl = pq[12 + jy0], wc = pq[8 + kc], omwc = pq[14 + kc];
      const T cmtodeg = sK[NANG * 4];
      const VT kk = *reinterpret_cast<const VT*>(sK + 4 * k);      // (double precision: two 16-byte reads)
      const T kk2 = sizeof(T) == 4 ? kk[2 % VEC] : sK[4 * k + 2], kk3 = sizeof(T) == 4 ? kk[3 % VEC] : sK[4 * k + 3];
#if ECWAM_HIP_CTU_STRICT
      T a2, b2, p2, m2;
      {
#pragma clang fp contract(off)
        const T tsp2 = tanph * kk[0], tsm2 = tanph * kk[1];
        a2 = m_max(tsp2, T(0)); p2 = m_max(-tsp2, T(0)); b2 = m_max(-tsm2, T(0)); m2 = m_max(tsm2, T(0));
      }
#else
      T ab2, p2, m2;
      ctu_fast_dir<T>(tanph, kk[0], kk[1], ab2, p2, m2);
      T ab2_lf = ab2, p2_lf = p2, m2_lf = m2;
      if constexpr (LF) ctu_fast_dir<T>(tanph, sK[NANG * 4 + 4 + 2 * k], sK[NANG * 4 + 4 + 2 * k + 1], ab2_lf, p2_lf, m2_lf);
#endif
      const T* bb = sB + q * 5 * NFRE + m;
      const VT bha = *reinterpret_cast<const VT*>(bb + jx0 * NFRE), bhb = *reinterpret_cast<const VT*>(bb + (1 - jx0) * NFRE),
               bya = *reinterpret_cast<const VT*>(bb + (2 + jy0) * NFRE), byb = *reinterpret_cast<const VT*>(bb + (3 - jy0) * NFRE),
               bc0 = *reinterpret_cast<const VT*>(bb + 4 * NFRE);
#if ECWAM_HIP_CTU_STRICT
      static_assert(!LF && !OBS, "the strict build of the weights has the plain form only");
#pragma unroll
      for (int i = 0; i < VEC; i += 2) {
#define P2(a) V2<T>{a[i], a[i + 1]}
        const V2<T> rr = ctu_w8_stencil_abs<T>(P2(bha), P2(bhb), P2(bya), P2(byb), P2(bc0), kk2, kk3, acpm1, zd, A.xdella, ga, A.delpro, cmtodeg, wl, omwl,
                                               wc, omwc, a2, b2, p2, m2, P2(b[0]), P2(b[1]), P2(b[2]), P2(b[3]), P2(b[4]), P2(b[5]), P2(b[6]), P2(b[7]));
#undef P2
        r[i] = rr.x; r[i + 1] = rr.y;
      }
#else
      const T zdg = pq[18], xdg = pq[19];
#pragma unroll
      for (int i = 0; i < VEC; i += 2) {
#define P2(a) V2<T>{a[i], a[i + 1]}
        const bool lf0 = LF && (m + i) < A.mlf, lf1 = LF && (m + i + 1) < A.mlf;
        CtuFastW8<T> w = ctu_fast_w8<T>(P2(bha), P2(bhb), P2(bya), P2(byb), P2(bc0), kk2, kk3, zd, A.xdella, ga, zdg, xdg, wl, omwl, wc, omwc,
                                        V2<T>{lf0 ? ab2_lf : ab2, lf1 ? ab2_lf : ab2}, V2<T>{lf0 ? p2_lf : p2, lf1 ? p2_lf : p2},
                                        V2<T>{lf0 ? m2_lf : m2, lf1 ? m2_lf : m2});
        if constexpr (OBS) {      // as k_propags2_otf: after the weights, before the stencil
          w.wlon = w.wlon * P2(b[8]); w.wlat1 = w.wlat1 * P2(b[9]); w.wlat2 = w.wlat2 * P2(b[9]); w.wcor1 = w.wcor1 * P2(b[10]); w.wcor2 = w.wcor2 * P2(b[10]);
        }
        const V2<T> rr = ctu_fast_apply<T>(w, P2(b[0]), P2(b[1]), P2(b[2]), P2(b[3]), P2(b[4]), P2(b[5]), P2(b[6]), P2(b[7]));
#undef P2
        r[i] = rr.x; r[i + 1] = rr.y;
      }
#endif
#pragma unroll
      for (int i = 0; i < VEC; i++)
        if (m + i < A.m0 || m + i >= A.m1) r[i] = b[0][i];      // outside the advected range: carried over
    }
    T* d = sT + m * RS + q * NANG + k;
#pragma unroll
    for (int i = 0; i < VEC; i++) d[i * RS] = r[i];
  };
  // the gathers of the first D - 1 steps leave before the direction-independent halves of the weights are built: their latency hides
  // behind the seven CGROUP gathers per (point, frequency) below
#pragma unroll
  for (int c = 0; c < D - 1 && c < NCH; c++) issue(c, buf[c % D]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr ((MODE & 1) != 0) {
    for (int i = lane; i < PP * NFRE; i += 64) {
      const int q = i / NFRE, m = i - q * NFRE;
      const int* iq = sI + q * 16;
      const T* pq = sPt + q * PTW;
      T cgl[2], cgy0[2], cgy1[2];
#pragma unroll
      for (int ic = 0; ic < 2; ic++) {
        cgl[ic] = A.cg[(size_t)iq[1 + ic] * NFRE + m];
        cgy0[ic] = A.cg[(size_t)iq[3 + 2 * ic] * NFRE + m];
        cgy1[ic] = A.cg[(size_t)iq[4 + 2 * ic] * NFRE + m];
      }
      const T wl[2] = {pq[6], pq[7]}, dp[2] = {pq[4], pq[5]};
      const CtuBase<T> b = ctu_base(A.cg[(size_t)iq[0] * NFRE + m], cgl, cgy0, cgy1, wl, dp);
      T* o = sB + q * 5 * NFRE + m;
#if ECWAM_HIP_CTU_STRICT
      o[0] = m_abs(b.h[0]); o[NFRE] = m_abs(b.h[1]); o[2 * NFRE] = m_abs(b.hy[0]); o[3 * NFRE] = m_abs(b.hy[1]); o[4 * NFRE] = b.cg0;
#else
      ctu_fast_planes<T>(b, pq[1], ((LF && m < A.mlf) ? A.delpro_lf : A.delpro) * sK[NANG * 4], o, o + NFRE, o + 2 * NFRE, o + 3 * NFRE);
      o[4 * NFRE] = b.cg0;
#endif
    }
    WSYNC();
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (c + D - 1 < NCH) issue(c + D - 1, buf[(c + D - 1) % D]);
    __builtin_amdgcn_sched_barrier(0);
    finish(c, buf[c % D]);
    __builtin_amdgcn_sched_barrier(0);
  }
#if defined(V4_ADV_PRIO)
  __builtin_amdgcn_s_setprio(0);
#endif
  WSYNC();
}

// TAUT_Z0 with the gravity-capillary roughness model (taut_z0.F90:148-287, LLGCBZ0 = T) for FOUR sea points at once: the 16 lanes of
// DPP row r work for one point (every argument is the value of the lane's point, the same on the lanes of a row), STRESS_GC's sum over
// the gravity-capillary wavenumbers NS .. NWAV_GC (stress_gc.F90:80-130, at most 82 terms) runs over the lanes of the row.  The
// iterations of a point stop by its own criterion (its lanes keep their values from then on); the wave leaves a loop when every row has.
// taut_z0_b_w (tests/csrc/implsch_v2.h) is the same arithmetic for one point per wave.
template <typename T>
__device__ T stress_gc_row(const DevTab<T>& tb, int l16, bool run, T ANG_GC, T USTAR, T Z0, T Z0MIN, T HALP, T RNFAC) {
  const T XLAMA = T(0.25), XLAMB = T(4.0);
  const int NS = ns_gc_d(tb, USTAR);
  const T t = USTAR * (Z0MIN / Z0);
  const T TAUWCG_MIN = t * t;
  const T XLAMBDA = T(1) + XLAMA * m_tanh(XLAMB * m_pow4(USTAR));
  const T LOGXL = m_log(XLAMBDA);
  const T hc = HALP * tb.C2OSQRTVG_GC[NS];
  const T ZABHRC = ANG_GC * tb.BETAMAXOXKAPPA2 * hc;
  const T CONST = tb.LLNORMAGAM ? RNFAC * tb.BMAXOKAP * hc / m_max(USTAR, tb.EPSUS) : T(0);
  T acc = T(0);
  for (int I = NS + l16;; I += 16) {
    const bool act = run && I <= tb.NWAV_GC;
    if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
    if (act) {
      const T X = USTAR * tb.CM_GC[I];
      const T XLOG = m_log(tb.XK_GC[I] * Z0) + tb.XKAPPA / (X + tb.ZALP);
      const T ZLOG = m_min(XLOG - LOGXL, T(0));
      const T ZLOG2X = ZLOG * ZLOG * X;
      const T GAM_W = ZLOG2X * ZLOG2X * m_exp(XLOG) * tb.OM3GMKM_GC[I];
      const T ZN = CONST * tb.XKMSQRTVGOC2_GC[I] * GAM_W;
      const T GAMNORMA = (T(1) + tb.RN1_RN * ZN) / (T(1) + ZN);
      const T wt = (I == NS) ? tb.DELKCC_GC_NS[NS] * tb.OMXKM3_GC[NS] : tb.DELKCC_OMXKM3_GC[I];
      acc = acc + (GAM_W * wt) * GAMNORMA;
    }
  }
  const T TAUWCG = v4_rowsum<T>(V2<T>{acc, T(0)}).x;
  return m_max(ZABHRC * TAUWCG, TAUWCG_MIN);
}
template <typename T>
__device__ void taut_z0_b_rows(const DevTab<T>& tb, int l16, int IUSFG, T HALP, T UTOP, T COSDIFF, T TAUW, T RNFAC, T& USTAR, T& Z0, T& Z0B,
                               T& CHRNCK) {
  const int NITER = 18;
  const T PMAX = T(0.99), Z0MIN = T(0.000001);
  const T US2TOTAUW = T(1) + tb.EPS1;
  const T RNUKAPPAM1 = (T(0.04) * tb.RNU) / tb.XKAPPA;
  const T PCE_GC = T(0.001) * IUSFG + (1 - IUSFG) * T(0.005);
  const T TAUWACT = m_max(TAUW * COSDIFF, tb.EPSMIN);
  const bool LLCOSDIFF = (COSDIFF > T(0.9));
  T ALPHAOG = T(0);
  if (tb.LLCAPCHNK) ALPHAOG = chnkmin(tb, UTOP) * tb.GM1;
  const T USMAX = m_max(-T(0.21339) + T(0.093698) * UTOP - T(0.0020944) * UTOP * UTOP + T(5.5091E-5) * UTOP * UTOP * UTOP, T(0.03));
  const T TAUWEFF = m_min(TAUWACT * US2TOTAUW, USMAX * USMAX);
  T X, CDFG;
  if (IUSFG == 0) {
    const T ALPHAGM1 = tb.ALPHA * tb.GM1;
    if (UTOP < T(1)) CDFG = T(0.002);
    else if (LLCOSDIFF) {
      const T um = m_max(USTAR, tb.EPSUS);
      X = m_min(TAUWACT / (um * um), PMAX);
      T ZCHAR = m_min(ALPHAGM1 * USTAR * USTAR / m_sqrt(T(1) - X), T(0.05) * m_exp(-T(0.05) * (UTOP - T(35.))));
      ZCHAR = m_min(ZCHAR, tb.ALPHAMAX);
      CDFG = tb.ACDLIN + tb.BCDLIN * m_sqrt(ZCHAR) * UTOP;
    } else CDFG = cdm_d(UTOP);
    USTAR = UTOP * m_sqrt(CDFG);
  }
  const T W1 = T(0.85) - T(0.05) * (m_tanh(T(10) * (UTOP - T(5))) + T(1));
  const T XKUTOP = tb.XKAPPA * UTOP;
  T USTOLD = USTAR;
  T TAUOLD = USTOLD * USTOLD;
  T TAUUNR = T(0);
  bool run = true;   // the point is still iterating; true after the loop = the reference's ITER > NITER
  for (int ITER = 1; ITER <= NITER; ITER++) {
    if (__builtin_amdgcn_ballot_w64(run) == 0ull) break;
    const T Z0n = m_max(tb.XNLEV / (m_exp(m_min(XKUTOP / USTOLD, T(50))) - T(1)), Z0MIN);
    const T TAUV = RNUKAPPAM1 * USTOLD / Z0n;
    const T ANG_GC = tb.ANG_GC_A + tb.ANG_GC_B * m_tanh(tb.ANG_GC_C * TAUOLD);
    const T TAUUNRn = stress_gc_row(tb, l16, run, ANG_GC, USTAR, Z0n, Z0MIN, HALP, RNFAC);
    const T TAUNEW = TAUWEFF + TAUV + TAUUNRn;
    const T USTNEW = m_sqrt(TAUNEW);
    const T USTARn = W1 * USTOLD + (T(1) - W1) * USTNEW;
    if (run) {
      Z0 = Z0n; TAUUNR = TAUUNRn; USTAR = USTARn;
      if (m_abs(USTAR - USTOLD) < PCE_GC * USTAR) run = false;
      else { TAUOLD = USTAR * USTAR; USTOLD = USTAR; }
    }
  }
  X = TAUWEFF / TAUOLD;
  if (run && X >= PMAX) {
    CDFG = cdm_d(UTOP);
    USTAR = UTOP * m_sqrt(CDFG);
    const T Z0MINRST = USTAR * USTAR * tb.ALPHA * tb.GM1;
    Z0 = m_max(tb.XNLEV / (m_exp(XKUTOP / USTAR) - T(1)), Z0MINRST);
    Z0B = Z0MINRST;
  } else {
    Z0 = m_max(tb.XNLEV / (m_exp(XKUTOP / USTAR) - T(1)), Z0MIN);
    Z0B = Z0 * m_sqrt(TAUUNR / TAUOLD);
  }
  if (X < PMAX) {   // Newton refinement: no lane exchange, every lane iterates for its point
    const T USNRF = USTAR, Z0NRF = Z0, Z0BNRF = Z0B;
    USTOLD = USTAR;
    TAUOLD = m_max(USTOLD * USTOLD, TAUWEFF);
    const T ALPOG = m_max(m_min(Z0B / TAUOLD, tb.ALPHAMAX), ALPHAOG);
    int ITER;
    for (ITER = 1; ITER <= NITER; ITER++) {
      X = m_min(TAUWEFF / TAUOLD, PMAX);
      const T USTM1 = T(1) / m_max(USTOLD, tb.EPSUS);
      const T Z0VIS = tb.RNUM * USTM1;
      const T HZ0VISO1MX = T(0.5) * Z0VIS / (T(1) - X);
      Z0B = ALPOG * TAUOLD;
      Z0 = HZ0VISO1MX + m_sqrt(HZ0VISO1MX * HZ0VISO1MX + Z0B * Z0B / (T(1) - X));
      const T XOLOGZ0 = T(1) / m_log(tb.XNLEV / Z0 + T(1));
      const T Fv = USTOLD - XKUTOP * XOLOGZ0;
      const T ZZ = T(2) * USTM1 * (T(3) * Z0B * Z0B + T(0.5) * Z0VIS * Z0 - Z0 * Z0) / (T(2) * Z0 * Z0 * (T(1) - X) - Z0VIS * Z0);
      const T DELF = T(1) - XKUTOP * XOLOGZ0 * XOLOGZ0 * ZZ;
      if (DELF != T(0)) USTAR = USTOLD - Fv / DELF;
      const T TAUNEW = m_max(USTAR * USTAR, TAUWEFF);
      USTAR = m_sqrt(TAUNEW);
      const T DEL = TAUNEW - TAUOLD;
      if (m_abs(DEL) < PCE_GC * TAUOLD) break;
      TAUOLD = TAUNEW;
      USTOLD = USTAR;
    }
    if (ITER > NITER) {
      USTAR = USNRF; Z0 = Z0NRF; Z0B = Z0BNRF;
      const T USTM1 = T(1) / m_max(USTAR, tb.EPSUS);
      const T Z0VIS = tb.RNUM * USTM1;
      CHRNCK = m_max(tb.G * (Z0 - Z0VIS) * USTM1 * USTM1, tb.ALPHAMIN);
    } else {
      const T um = m_max(USTAR, tb.EPSUS);
      CHRNCK = m_max(tb.G * (Z0B / m_sqrt(T(1) - X)) / (um * um), tb.ALPHAMIN);
    }
  } else {
    const T USTM1 = T(1) / m_max(USTAR, tb.EPSUS);
    const T Z0VIS = tb.RNUM * USTM1;
    CHRNCK = m_max(tb.G * (Z0 - Z0VIS) * USTM1 * USTM1, tb.ALPHAMIN);
  }
}

// SINPUT_ARD (sinput_ard.F90:153-520) for one SINFLX call.  Outputs: XLLWS masks of the two directions of the lane (bit m), the row
// integrals X, Y of the frequencies the lane owns (m = s G + j), the FEMEANWS integrands (wse: x = SUM DFIM F, y = SUM DFIMOFR F
// over the windsea bins; wslast = windsea part of the last row), apl (negative wind input per direction) and -- LLSNEG -- the
// wind-input coefficient of every row into gfl (the point's XLLWS block, [M][K]).
template <typename T, int NANG, int PP, int NGST, bool LLSNEG>
__device__ void v4_sinput(const DevTab<T>& tb, const V4Ctx<T, NANG, PP>& L, T UFRIC, T Z0M, T RAORW, T SIG_N, T TEMP2, T PTURB, T AIRD_PVISC,
                          T sinwd, T coswd, T* __restrict__ gfl, T* __restrict__ gsp, unsigned long long& xm0, unsigned long long& xm1, V2<T>& wse,
                          V2<T>& wslast, V2<T>& apl, T (&rX)[V4_NS(NANG)], T (&rY)[V4_NS(NANG)], T* __restrict__ sXY) {
  constexpr int G = NANG / 2, NFRE = V4_NFRE, RS = PP * NANG, NS = V4_NS(NANG);
  const T CONST1 = tb.BETAMAXOXKAPPA2, ABS_TAUWSHELTER = m_abs(tb.TAUWSHELTER);
  const T FU = m_abs(tb.SWELLF3), FUD = tb.SWELLF2, ROGOROAIR = tb.G / RAORW;
  const T AVG = T(1) / T(NGST);
  for (int m = L.j; m < NFRE; m += G) L.zcn[m] = m_log(L.fac4[m * 4 + Q4_WAVNUM] * Z0M);
  WSYNC();
  const T XKAPPA = tb.XKAPPA, ZALP = tb.ZALP;
  // the two gust states as the halves of a pair (NGST = 1: both halves the same state): the wave-uniform chain of a row -- sheltered
  // stress, its direction and magnitude, U*/c -- is packed arithmetic on them
  V2<T> vUSTP = (NGST == 1) ? V2<T>{UFRIC, UFRIC} : V2<T>{UFRIC * (T(1) + SIG_N), UFRIC * (T(1) - SIG_N)};
  V2<T> vXS = {T(0), T(0)}, vYS = {T(0), T(0)};
  const V2<T> vTAUX = (vUSTP * vUSTP) * sinwd, vTAUY = (vUSTP * vUSTP) * coswd;
  xm0 = 0ull; xm1 = 0ull;
  const V2<T> z2 = {T(0), T(0)};
  wse = z2; wslast = z2; apl = z2;
#pragma unroll
  for (int s = 0; s < NS; s++) { rX[s] = T(0); rY[s] = T(0); }
  const T* tF = L.tile + L.own;
  // operands of the next row are read one row ahead (the row itself, CINV / WAVNUM, LOG(WAVNUM Z0M))
  V2<T> f_n = *reinterpret_cast<const V2<T>*>(tF);
  V2<T> cw_n = *reinterpret_cast<const V2<T>*>(L.fac4 + Q4_CINV);
  T zcn_n = L.zcn[0];
  // V4_RECPF: the row's record of module constants one row ahead as well (its scalar load would otherwise be waited for where it is issued)
  constexpr bool RPF = (V4_RECPF != 0) && sizeof(T) == 4;
  T rw_n[6];
  if constexpr (RPF) {
#pragma unroll
    for (int i = 0; i < 6; i++) rw_n[i] = tb.SINROW[0][i];
  }
#pragma unroll 2
  for (int m = 0; m < NFRE; m++) {
    const V2<T> f = f_n, cw = cw_n;
    const T ZCN = zcn_n, cinv_m = cw.x;
    T rw[6];
    if constexpr (RPF) {
#pragma unroll
      for (int i = 0; i < 6; i++) rw[i] = rw_n[i];
    }
    {
      const int mn = m + 1 < NFRE ? m + 1 : m;
      f_n = *reinterpret_cast<const V2<T>*>(tF + mn * RS);
      cw_n = *reinterpret_cast<const V2<T>*>(L.fac4 + mn * 4 + Q4_CINV);
      zcn_n = L.zcn[mn];
      if constexpr (RPF) {
#pragma unroll
        for (int i = 0; i < 6; i++) rw_n[i] = tb.SINROW[mn][i];
      }
    }
    if constexpr (RPF) __builtin_amdgcn_sched_barrier(0);      // (left to the scheduler the loads of the next row sink to the end of this one)
    V4_CHK(m >= 0 && m < NFRE);
    const T* row = RPF ? rw : tb.SINROW[m];   // one scalar load: ZPIFR, DFIM, C5, T1, RHOWG_DFIM, DFIMOFR of the row
    const T SIGm = row[0], DFIMm = row[1];
    const T CONSTF = ROGOROAIR * cinv_m * DFIMm;
    const T DSTAB1 = LLSNEG ? (row[2] * AIRD_PVISC) * cw.y : T(0);
    const T CNSN = (SIGm * CONST1) * RAORW;
    const T TEMP1 = LLSNEG ? row[3] * RAORW : T(0);
    V2<T> SLP[2], FLP[2];
    bool xl0 = false, xl1 = false;
    const V2<T> vTPX = vTAUX - ABS_TAUWSHELTER * vXS, vTPY = vTAUY - ABS_TAUWSHELTER * vYS;
    const V2<T> vh2 = vTPX * vTPX + vTPY * vTPY;
    // |TAUP| = h2 / SQRT(h2) with the root of MAX(h2, tiny): a vanishing stress gives h = 0 exactly and a finite direction (0, 0) --
    // which direction does not matter then, every term it enters carries the factor USTP = 0 -- without the selects of a zero test
    const T TINY = sizeof(T) == 4 ? T(1e-36) : T(1e-300);
    V2<T> vrh = {fs_rsq<8>(m_max(vh2.x, TINY)), (NGST == 2) ? fs_rsq<8>(m_max(vh2.y, TINY)) : T(0)};
    if (NGST == 1) vrh.y = vrh.x;
    const V2<T> vh = vh2 * vrh, vCOSU = vTPY * vrh, vSINU = vTPX * vrh;
    vUSTP = V2<T>{fs_sqrt<8>(vh.x), (NGST == 2) ? fs_sqrt<8>(vh.y) : T(0)};
    if (NGST == 1) vUSTP.y = vUSTP.x;
    const V2<T> vUCN = vUSTP * cinv_m;
    const V2<T> vden = vUCN + ZALP;
    const V2<T> vUZ = XKAPPA * V2<T>{fs_rcp<8>(vden.x), (NGST == 2) ? fs_rcp<8>(vden.y) : T(0)};
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      const T COSU = ig ? vCOSU.y : vCOSU.x, SINU = ig ? vSINU.y : vSINU.x;
      const T UCN = ig ? vUCN.y : vUCN.x, UCNZALPD = ig ? vUZ.y : vUZ.x;
      const T USTPg = ig ? vUSTP.y : vUSTP.x;
      const V2<T> coslp = L.costh * COSU + L.sinth * SINU;
      V2<T> gam0 = z2;
      {
        const bool c0 = coslp.x > T(0.01), c1 = coslp.y > T(0.01);
        const T Z0 = ZCN + UCNZALPD * fs_rcp<4>(coslp.x), Z1 = ZCN + UCNZALPD * fs_rcp<4>(coslp.y);
        const bool n0 = c0 && (Z0 < T(0)), n1 = c1 && (Z1 < T(0));
        // single precision: no test for "no lane grows" -- a uniform branch here costs more than the two v_exp_f32 it would skip;
        // the double-precision exponentials are long instruction sequences worth skipping
        if (sizeof(T) == 4 || __builtin_amdgcn_ballot_w64(n0 || n1) != 0ull) {
          const V2<T> ZL = {Z0, Z1};
          const V2<T> Z2X = ZL * ZL * (coslp * UCN);
          const V2<T> ex = {f_exp(Z0), f_exp(Z1)};
          const V2<T> g = ex * Z2X * Z2X * CNSN;
          gam0.x = n0 ? g.x : T(0);
          gam0.y = n1 ? g.y : T(0);
          xl0 = xl0 || n0;
          xl1 = xl1 || n1;
        }
      }
      V2<T> dstab = z2;
      if (LLSNEG) {
        const V2<T> DSTAB2 = TEMP1 * (TEMP2 + (FU + FUD * coslp) * USTPg);
        dstab = DSTAB1 + PTURB * DSTAB2;
      }
      FLP[ig] = gam0 + dstab;
      SLP[ig] = gam0 * f;
    }
    V2<T> sp = SLP[0], fl = FLP[0];
    if (NGST == 2) { sp = sp + SLP[1]; fl = fl + FLP[1]; }
    sp = AVG * sp;
    fl = AVG * fl;
    T xrow = T(0), yrow = T(0);
    // V4_REDN (single precision, where the branch above is always taken): the all-reduces of the row -- the stress of each gust state and, in
    // the second call, the row's positive input -- in one batch: two LDS round trips per row instead of five
    constexpr bool REDN = (V4_REDN != 0) && sizeof(T) == 4;
    if constexpr (REDN) {
      constexpr int NR = 2 * NGST + (LLSNEG ? 1 : 0);
      T red[NR];
#pragma unroll
      for (int ig = 0; ig < NGST; ig++) {
        const V2<T> sx = SLP[ig] * L.sinth, sy = SLP[ig] * L.costh;
        red[2 * ig] = sx.x + sx.y; red[2 * ig + 1] = sy.x + sy.y;
      }
      if constexpr (LLSNEG) red[2 * NGST] = sp.x + sp.y;
      v4_allsum_n<G, NR, (G == 18 ? 2 * NGST : NR)>(red, L.rot);
#pragma unroll
      for (int ig = 0; ig < NGST; ig++) {
        if (ig == 0) { vXS.x = vXS.x + CONSTF * red[0]; vYS.x = vYS.x + CONSTF * red[1]; }
        else { vXS.y = vXS.y + CONSTF * red[2 * ig]; vYS.y = vYS.y + CONSTF * red[2 * ig + 1]; }
        xrow += red[2 * ig];
        yrow += red[2 * ig + 1];
      }
      xrow = AVG * xrow; yrow = AVG * yrow;
      if constexpr (LLSNEG) gsp[m] = red[2 * NGST];      // (v4_row_total_to_lds: every lane of the row stores the same total)
    } else
    if (sizeof(T) == 4 || __builtin_amdgcn_ballot_w64(xl0 || xl1) != 0ull) {
      {
#pragma unroll
      for (int ig = 0; ig < NGST; ig++) {
        const V2<T> sx = SLP[ig] * L.sinth, sy = SLP[ig] * L.costh;
        const V2<T> xs = v4_allsum<G, T>(V2<T>{sx.x + sx.y, sy.x + sy.y}, L.rot);
        if (ig == 0) { vXS.x = vXS.x + CONSTF * xs.x; vYS.x = vYS.x + CONSTF * xs.y; }
        else { vXS.y = vXS.y + CONSTF * xs.x; vYS.y = vYS.y + CONSTF * xs.y; }
        xrow += xs.x;
        yrow += xs.y;
      }
      }
      xrow = AVG * xrow; yrow = AVG * yrow;
    }
    if constexpr (G == 18) {
      // the row integrals into the (idle) staging rows: the symmetric rotations of the all-reduce leave the same bits on every lane of
      // the point, so all of them store
      *reinterpret_cast<V2<T>*>(sXY + 2 * m) = V2<T>{xrow, yrow};
    } else {  // (sums of three rotated terms differ in the last bit between lanes) the lane that owns frequency m keeps its row integrals
      const int ms = m / G, mj = m - ms * G;
      const bool mine = (L.j == mj);
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const bool w = mine && (ms == s);
        rX[s] = w ? xrow : rX[s];
        rY[s] = w ? yrow : rY[s];
      }
    }
    if (LLSNEG) {
      apl = apl + (fl * f - sp) * row[4];
      *reinterpret_cast<V2<T>*>(gfl + (size_t)m * NANG) = fl;
      // the row's positive input summed over the directions -> the point's row table (weighted below the cut-off once MIJ is known)
      if constexpr (!REDN) v4_row_total_to_lds<G, T>(sp.x + sp.y, L.rot, gsp, m);
    }
    if (xl0) xm0 |= (1ull << m);
    if (xl1) xm1 |= (1ull << m);
    const V2<T> x = {xl0 ? f.x : T(0), xl1 ? f.y : T(0)};
    wse = wse + V2<T>{DFIMm, row[5]} * (x.x + x.y);
    wslast = x;
    if constexpr (RPF) v4_pin6(rw_n);
  }
  WSYNC();
}

// SINPUT_ARD with the normalised growth rate (LLNORMAGAM = T, sinput_ard.F90:380-437; TAUWSHELTER = 0 in that physics: no sheltering
// recurrence, the growth direction is the wind direction).  Per row and gust state one all-reduce for SUMF / SUMFSIN2, one for the row
// integrals.  xng: plane [M] of CONSTN RNFAC / RAORW XK2CG(M) (filled here, in the plane SQRT(WAVNUM) leaves free between the two FKMEAN).
template <typename T, int NANG, int PP, int NGST, bool LLSNEG>
__device__ void v4_sinput_n(const DevTab<T>& tb, const V4Ctx<T, NANG, PP>& L, const T* __restrict__ xk2cg, T UFRIC, T Z0M, T RAORW, T RNFAC,
                            T SIG_N, T TEMP2, T PTURB, T AIRD_PVISC, V2<T> coswdif, V2<T> sinwdif2, T* __restrict__ gfl, T* __restrict__ gsp,
                            unsigned long long& xm0, unsigned long long& xm1, V2<T>& wse, V2<T>& wslast, V2<T>& apl,
                            T (&rX)[V4_NS(NANG)], T (&rY)[V4_NS(NANG)], T* __restrict__ sXY) {
  constexpr int G = NANG / 2, NFRE = V4_NFRE, RS = PP * NANG, NS = V4_NS(NANG);
  const T CONST1 = tb.BETAMAXOXKAPPA2;
  const T CSTRNFAC = (tb.DELTH / (tb.XKAPPA * tb.ZPI)) * RNFAC / RAORW;
  const T FU = m_abs(tb.SWELLF3), FUD = tb.SWELLF2;
  const T AVG = T(1) / T(NGST);
  T* xng = L.sq;
  for (int m = L.j; m < NFRE; m += G) {
    L.zcn[m] = m_log(L.fac4[m * 4 + Q4_WAVNUM] * Z0M);
    xng[m] = CSTRNFAC * xk2cg[m];
  }
  WSYNC();
  const T XKAPPA = tb.XKAPPA, ZALP = tb.ZALP;
  T USTP[2], USTPM1[2];
  if (NGST == 1) USTP[0] = UFRIC;
  else { USTP[0] = UFRIC * (T(1) + SIG_N); USTP[1] = UFRIC * (T(1) - SIG_N); }
#pragma unroll
  for (int ig = 0; ig < NGST; ig++) USTPM1[ig] = T(1) / m_max(USTP[ig], tb.EPSUS);
  xm0 = 0ull; xm1 = 0ull;
  const V2<T> z2 = {T(0), T(0)};
  wse = z2; wslast = z2; apl = z2;
#pragma unroll
  for (int s = 0; s < NS; s++) { rX[s] = T(0); rY[s] = T(0); }
  const bool c0 = coswdif.x > T(0.01), c1 = coswdif.y > T(0.01);
  const V2<T> rcos = {f_rcp(coswdif.x), f_rcp(coswdif.y)};
  T GAMNORMA[2] = {T(1), T(1)};
  const T* tF = L.tile + L.own;
  constexpr bool RPF = (V4_RECPF != 0) && sizeof(T) == 4;      // the next row's module constants fetched ahead (see v4_sinput)
  T rw_n[6];
  if constexpr (RPF) {
#pragma unroll
    for (int i = 0; i < 6; i++) rw_n[i] = tb.SINROW[0][i];
  }
#pragma unroll 2
  for (int m = 0; m < NFRE; m++) {
    T rw[6];
    if constexpr (RPF) {
#pragma unroll
      for (int i = 0; i < 6; i++) rw[i] = rw_n[i];
      const int mn = m + 1 < NFRE ? m + 1 : m;
#pragma unroll
      for (int i = 0; i < 6; i++) rw_n[i] = tb.SINROW[mn][i];
      __builtin_amdgcn_sched_barrier(0);
    }
    const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
    const V2<T> cw = *reinterpret_cast<const V2<T>*>(L.fac4 + m * 4 + Q4_CINV);
    const T ZCN = L.zcn[m], cinv_m = cw.x, XNGAMCONST = xng[m];
    const T* row = RPF ? rw : tb.SINROW[m];   // one scalar load: ZPIFR, DFIM, C5, T1, RHOWG_DFIM, DFIMOFR of the row
    const T SIGm = row[0], DFIMm = row[1];
    const T DSTAB1 = LLSNEG ? (row[2] * AIRD_PVISC) * cw.y : T(0);
    const T CNSN = (SIGm * CONST1) * RAORW;
    const T TEMP1 = LLSNEG ? row[3] * RAORW : T(0);
    V2<T> SLP[2], FLP[2];
    bool xl0 = false, xl1 = false;
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      const T UCN = USTP[ig] * cinv_m;
      const T UCNZALPD = XKAPPA * f_rcp(UCN + ZALP);
      V2<T> gam0 = z2;
      const T Z0 = ZCN + UCNZALPD * rcos.x, Z1 = ZCN + UCNZALPD * rcos.y;
      const bool n0 = c0 && (Z0 < T(0)), n1 = c1 && (Z1 < T(0));
      if (__builtin_amdgcn_ballot_w64(n0 || n1) != 0ull) {
        const V2<T> ZL = {Z0, Z1};
        const V2<T> Z2X = ZL * ZL * (coswdif * UCN);
        const V2<T> ex = {f_exp(Z0), f_exp(Z1)};
        const V2<T> g = ex * Z2X * Z2X * CNSN;
        gam0.x = n0 ? g.x : T(0);
        gam0.y = n1 ? g.y : T(0);
        xl0 = xl0 || n0;
        xl1 = xl1 || n1;
        const V2<T> a = gam0 * f, as2 = a * sinwdif2;
        const V2<T> sm = v4_allsum<G, T>(V2<T>{a.x + a.y, as2.x + as2.y}, L.rot);   // SUMF, SUMFSIN2
        const T ZNZ = XNGAMCONST * USTPM1[ig];
        GAMNORMA[ig] = (T(1) + ZNZ * sm.y) / (T(1) + ZNZ * sm.x);
      }
      V2<T> dstab = z2;
      if (LLSNEG) {
        const V2<T> DSTAB2 = TEMP1 * (TEMP2 + (FU + FUD * coswdif) * USTP[ig]);
        dstab = DSTAB1 + PTURB * DSTAB2;
      }
      const V2<T> gn = gam0 * GAMNORMA[ig];
      FLP[ig] = gn + dstab;
      SLP[ig] = gn * f;
    }
    V2<T> sp = SLP[0], fl = FLP[0];
    if (NGST == 2) { sp = sp + SLP[1]; fl = fl + FLP[1]; }
    sp = AVG * sp;
    fl = AVG * fl;
    const bool anygrow = __builtin_amdgcn_ballot_w64(xl0 || xl1) != 0ull;
    T xrow = T(0), yrow = T(0);
    if (anygrow) {
      const V2<T> sx = sp * L.sinth, sy = sp * L.costh;
      const V2<T> xs = v4_allsum<G, T>(V2<T>{sx.x + sx.y, sy.x + sy.y}, L.rot);
      xrow = xs.x; yrow = xs.y;
    }
    if constexpr (G == 18) {
      // the row integrals into the (idle) staging rows: the symmetric rotations of the all-reduce leave the same bits on every lane of
      // the point, so all of them store
      *reinterpret_cast<V2<T>*>(sXY + 2 * m) = V2<T>{xrow, yrow};
    } else {  // (sums of three rotated terms differ in the last bit between lanes) the lane that owns frequency m keeps its row integrals
      const int ms = m / G, mj = m - ms * G;
      const bool mine = (L.j == mj);
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const bool w = mine && (ms == s);
        rX[s] = w ? xrow : rX[s];
        rY[s] = w ? yrow : rY[s];
      }
    }
    if (LLSNEG) {
      apl = apl + (fl * f - sp) * row[4];
      *reinterpret_cast<V2<T>*>(gfl + (size_t)m * NANG) = fl;
      v4_row_total_to_lds<G, T>(sp.x + sp.y, L.rot, gsp, m);
    }
    if (xl0) xm0 |= (1ull << m);
    if (xl1) xm1 |= (1ull << m);
    const V2<T> x = {xl0 ? f.x : T(0), xl1 ? f.y : T(0)};
    wse = wse + V2<T>{DFIMm, row[5]} * (x.x + x.y);
    wslast = x;
    if constexpr (RPF) v4_pin6(rw_n);
  }
  WSYNC();
}

// SINPUT_JAN (sinput_jan.F90:171-396, IPHYS = 0): Janssen's wind input with gustiness and the swell damping of IDAMPING.
// No sheltering: the rows do not depend on each other (the row integrals of the stress are still reduced row by row, but nothing
// waits for them).  Same outputs as v4_sinput.  coswdif: COS(TH - WDWAVE) of the lane's pair.  NRM (the builds that carry LLNORMAGAM,
// decided at run time by norma): the growth rate of a row and gust state renormalised by GAMNORMA = (1 + ZNZ SUMFSIN2) / (1 + ZNZ SUMF)
// (sinput_jan.F90:329-357; one all-reduce per row and gust state); xng: plane [M] of CONSTN RNFAC / RAORW XK2CG(M), filled here.
template <typename T, int NANG, int PP, int NGST, bool LLSNEG, bool NRM = false>
__device__ void v4_sinput_jan(const DevTab<T>& tb, const V4Ctx<T, NANG, PP>& L, T UFRIC, T Z0M, T RAORW, T SIG_N, V2<T> coswdif,
                              T* __restrict__ gfl, T* __restrict__ gsp, unsigned long long& xm0, unsigned long long& xm1, V2<T>& wse,
                              V2<T>& wslast, V2<T>& apl, T (&rX)[V4_NS(NANG)], T (&rY)[V4_NS(NANG)], T* __restrict__ sXY,
                              bool norma = false, V2<T> sinwdif2 = V2<T>{T(0), T(0)}, const T* __restrict__ xk2cg = nullptr, T RNFAC = T(1)) {
  constexpr int G = NANG / 2, NFRE = V4_NFRE, RS = PP * NANG, NS = V4_NS(NANG);
  const T CONST1 = tb.BETAMAXOXKAPPA2;
  const T CONST3 = T(tb.IDAMPING) * (T(2) * tb.XKAPPA / CONST1);
  const T XKAPPAD = T(1) / tb.XKAPPA;
  T* xng = L.sq;
  for (int m = L.j; m < NFRE; m += G) {
    L.zcn[m] = m_log(L.fac4[m * 4 + Q4_WAVNUM] * Z0M);
    if constexpr (NRM) {
      if (norma) xng[m] = ((tb.DELTH / (tb.XKAPPA * tb.ZPI)) * RNFAC / RAORW) * xk2cg[m];
    }
  }
  WSYNC();
  // gust states (sinput_jan.F90:200-246): US = UFRIC (1 -+ SIG_N), weights 1/2 each
  const T WS = T(1) / T(NGST);
  const V2<T> vUS = (NGST == 1) ? V2<T>{UFRIC, UFRIC} : V2<T>{UFRIC * (T(1) - SIG_N), UFRIC * (T(1) + SIG_N)};
  const V2<T> vUSM1 = {T(1) / m_max(vUS.x, tb.EPSUS), T(1) / m_max(vUS.y, tb.EPSUS)};
  const bool c0 = coswdif.x > T(0.01), c1 = coswdif.y > T(0.01);
  const V2<T> xkoc = {tb.XKAPPA * f_rcp(coswdif.x), tb.XKAPPA * f_rcp(coswdif.y)};
  xm0 = 0ull; xm1 = 0ull;
  const V2<T> z2 = {T(0), T(0)};
  wse = z2; wslast = z2; apl = z2;
#pragma unroll
  for (int s = 0; s < NS; s++) { rX[s] = T(0); rY[s] = T(0); }
  const T* tF = L.tile + L.own;
  constexpr bool RPF = (V4_RECPF != 0) && sizeof(T) == 4;      // the next row's module constants fetched ahead (see v4_sinput)
  T rw_n[6];
  if constexpr (RPF) {
#pragma unroll
    for (int i = 0; i < 6; i++) rw_n[i] = tb.SINROW[0][i];
  }
#pragma unroll 2
  for (int m = 0; m < NFRE; m++) {
    T rw[6];
    if constexpr (RPF) {
#pragma unroll
      for (int i = 0; i < 6; i++) rw[i] = rw_n[i];
      const int mn = m + 1 < NFRE ? m + 1 : m;
#pragma unroll
      for (int i = 0; i < 6; i++) rw_n[i] = tb.SINROW[mn][i];
      __builtin_amdgcn_sched_barrier(0);
    }
    const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
    const V2<T> cw = *reinterpret_cast<const V2<T>*>(L.fac4 + m * 4 + Q4_CINV);   // CINV, WAVNUM
    const T ZCN = L.zcn[m], cinv_m = cw.x;
    const T* row = RPF ? rw : tb.SINROW[m];   // ZPIFR, DFIM, -, -, RHOWG_DFIM, DFIMOFR of the row
    const T SIGm = row[0], DFIMm = row[1];
    const T ZTANHKD = (SIGm * SIGm) * f_rcp(tb.G * cw.y);
    const T CNSN = (SIGm * CONST1) * ZTANHKD * RAORW;
    const V2<T> vUCN = vUS * cinv_m + tb.ZALP;
    const V2<T> vUCND = {f_rcp(vUCN.x), (NGST == 2) ? f_rcp(vUCN.y) : T(0)};
    V2<T> ufac1 = z2, ufac2 = z2;
    bool xl0 = false, xl1 = false;
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      const T UCN = ig ? vUCN.y : vUCN.x, UCND = ig ? vUCND.y : vUCND.x, US = ig ? vUS.y : vUS.x;
      const T Z0 = ZCN + xkoc.x * UCND, Z1 = ZCN + xkoc.y * UCND;
      const bool n0 = c0 && (Z0 < T(0)), n1 = c1 && (Z1 < T(0));
      if (sizeof(T) == 4 || __builtin_amdgcn_ballot_w64(n0 || n1) != 0ull) {
        const V2<T> ZL = {Z0, Z1};
        const V2<T> Z2X = ZL * ZL * (coswdif * UCN);
        const V2<T> ex = {f_exp(Z0), f_exp(Z1)};
        const V2<T> g = ex * Z2X * Z2X * CNSN;
        V2<T> g0 = {n0 ? g.x : T(0), n1 ? g.y : T(0)};      // GAM0 of the gust state
        if constexpr (NRM) {
          if (norma) {
            const V2<T> a = g0 * f, as2 = a * sinwdif2;
            const V2<T> sm = v4_allsum<G, T>(V2<T>{a.x + a.y, as2.x + as2.y}, L.rot);   // SUMF, SUMFSIN2
            const T ZNZ = xng[m] * (ig ? vUSM1.y : vUSM1.x);
            g0 = g0 * ((T(1) + ZNZ * sm.y) / (T(1) + ZNZ * sm.x));
          }
        }
        ufac1 = ufac1 + WS * g0;
        xl0 = xl0 || n0;
        xl1 = xl1 || n1;
      }
      if (LLSNEG) {   // swell damping (sinput_jan.F90:368-381)
        const T C3 = CONST3 * (UCN * UCN);
        const T XVD = f_rcp(-US * XKAPPAD * ZCN * cinv_m);
        ufac2 = ufac2 + (WS * C3) * (coswdif - XVD);
      }
    }
    const V2<T> fl = ufac1 + ufac2 * CNSN;
    const V2<T> sp = ufac1 * f;
    T xrow = T(0), yrow = T(0);
    if (sizeof(T) == 4 || __builtin_amdgcn_ballot_w64(xl0 || xl1) != 0ull) {
      const V2<T> sx = sp * L.sinth, sy = sp * L.costh;
      const V2<T> xs = v4_allsum<G, T>(V2<T>{sx.x + sx.y, sy.x + sy.y}, L.rot);
      xrow = xs.x; yrow = xs.y;
    }
    if constexpr (G == 18) {
      *reinterpret_cast<V2<T>*>(sXY + 2 * m) = V2<T>{xrow, yrow};
    } else {
      const int ms = m / G, mj = m - ms * G;
      const bool mine = (L.j == mj);
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const bool w = mine && (ms == s);
        rX[s] = w ? xrow : rX[s];
        rY[s] = w ? yrow : rY[s];
      }
    }
    if (LLSNEG) {
      apl = apl + (fl * f - sp) * row[4];
      *reinterpret_cast<V2<T>*>(gfl + (size_t)m * NANG) = fl;
      v4_row_total_to_lds<G, T>(sp.x + sp.y, L.rot, gsp, m);
    }
    if (xl0) xm0 |= (1ull << m);
    if (xl1) xm1 |= (1ull << m);
    const V2<T> x = {xl0 ? f.x : T(0), xl1 ? f.y : T(0)};
    wse = wse + V2<T>{DFIMm, row[5]} * (x.x + x.y);
    wslast = x;
    if constexpr (RPF) v4_pin6(rw_n);
  }
  WSYNC();
}

// Attenuation rates of the sea ice per frequency (without CGROUP and the ice cover): SDICE1's scattering ALP = EXP(CIDEAC(period, thickness))
// DINV ZALPFACB, CIDEAC bilinear (sdice1.F90:135-168), and SDICE3's viscous friction (sdice3.F90:117-137).  One definition for the factor
// table of the point and for the RARE build's SLICE.
template <typename T>
__device__ __forceinline__ T v4_sdice1_alp(const DevTab<T>& tb, int m, T CITHICKv, T DINV) {
  const int NICT = tb.NICT, NICH = tb.NICH;
  const T TW = T(1) / tb.FR[m];
  int IT = (int)m_floor((TW - tb.TICMIN) / tb.DTIC + T(1));
  IT = IT < 1 ? 1 : (IT > NICT ? NICT : IT);
  const int IT1 = IT + 1 > NICT ? NICT : IT + 1;
  const T WT1 = m_max(m_min(T(1), (TW - (tb.TICMIN + T(IT - 1) * tb.DTIC)) / tb.DTIC), T(0));
  const T WT = T(1) - WT1;
  int IH = (int)m_floor((CITHICKv - tb.HICMIN) / tb.DHIC + T(1));
  IH = IH < 1 ? 1 : (IH > NICH ? NICH : IH);
  const int IH1 = IH + 1 > NICH ? NICH : IH + 1;
  const T WH1 = m_max(m_min(T(1), (CITHICKv - (tb.HICMIN + T(IH - 1) * tb.DHIC)) / tb.DHIC), T(0));
  const T WH = T(1) - WH1;
  const T* cd = tb.CIDEAC;
  const T CI = WT * (WH * cd[(IH - 1) * NICT + IT - 1] + WH1 * cd[(IH1 - 1) * NICT + IT - 1]) +
               WT1 * (WH * cd[(IH - 1) * NICT + IT1 - 1] + WH1 * cd[(IH1 - 1) * NICT + IT1 - 1]);
  return m_exp(CI) * DINV * tb.ZALPFACB;
}
template <typename T>
__device__ __forceinline__ T v4_sdice3_alp(const DevTab<T>& tb, int m, T CITHICKv, T ALPFAC) {
  const T CDICE = T(0.1274) * m_pow(tb.ZPI / m_sqrt(tb.G), T(4.5));
  return (T(2) * CDICE * m_pow(CITHICKv, T(1.25)) * m_pow(tb.FR[m], T(4.5))) * ALPFAC;
}

// One wavefront advances PP sea points.  Lanes beyond PP G shadow other lanes and the points of a short last wave shadow its
// last point: shadows run the same instructions on the same data, so their LDS and global stores repeat their original's values
// at the same addresses -- no store is predicated.
// EXT: the build that also carries LLGCBZ0 (gravity-capillary roughness: HALPHAP, TAUT_Z0 with STRESS_GC per point across the wave)
// and LLNORMAGAM (normalised growth rate) -- the cy49r1 / cy50r1 physics; the flag-set-A build has none of that code.
// JAN: IPHYS = 0 (sinput_jan.F90 + sdissip_jan.F90: the dissipation is a rate per (point, frequency) in the saturation slot of the
// factor table; no saturation filter, no sheltering recurrence).  ENHMC: ISNONLIN = 1 (the DIA scaled per interaction frequency by
// TRANSF(k(MC), DEPTH), snonlin.F90:138-150).
// RARE: the build for what no registered configuration selects, decided at run time inside it: ISNONLIN = 2 (TRANSF_SNL with the spectral
// widths of PEAK_ANG, snonlin.F90:152-165; ENHMC builds only), LCIWA2 (sdice2.F90: the attenuation depends on the bin's own energy), the
// ice radiative stress LWNEMOCOUWRS (wnfluxes.F90:178-196) and strain LWNEMOCOUSTRN (cimsstrn.F90), friction-velocity forcing ICODE = 1, 2
// (airsea.F90:100-117) and LWVFLX_SNL = F (implsch.F90:280-288) -- uniform branches, which the common builds do not pay for.
// PART: 0 = the whole time step in one kernel (the product path of every build but the double precision RARE ones); 1 / 2 = the two-kernel
// split of round 5: PART 1 runs the prologue, SDEPTHLIM / FKMEAN, both SINFLX calls and the scalar chain between
// them, writes XLLWS, parks the wind-input coefficient in the rows of wi[ij][M][K] (context-owned) and hands its scalars over in fin;
// PART 2 loads the spectrum again, re-applies SDEPTHLIM's scale and tail (the same operations: the same bits) and runs the sweep, the
// fluxes, the tail and the stores.  Same source, same results bit for bit; two smaller functions for the compiler.
// ADV: 0 = the tile is loaded from FL1 (IMPLSCH on its own, behind a PROPAGS2 kernel); 1 = the tile load IS the advection (round 6:
// v4_advect_tile above -- PROPAGS2 of the wave's points from the rows of adv.f_in straight into the tile, the new spectrum stored to the rows
// of fl1, which must be another buffer: one kernel per WAMINTGR step); 3 = the same for the native O1280 mode: the fast waves with their own
// time step, read from the compact rows their sub-steps left (V4Adv::gin); + 4 (5, 7) = with the sub-grid obstructions of LSUBGRID (V4Adv::obs);
// 2 = the go / no-go probe of ADV = 1 (the eight gathers of the stencil through the real neighbour tables, made-up weights).
template <typename T, int NANG, int PP, int R1, int R2, int NH, bool EXT, bool JAN = false, bool ENHMC = false, bool RARE = false, int PART = 0, int ADV = 0>
// single precision: two waves per SIMD (LDS: 8 waves per CU), at most 256 VGPRs; double precision: the LDS holds one wave per SIMD
// (40 KB per wave) and the kernel may use the whole register file (340 registers: no scratch)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(V4_WPE_MIN(T), 2)))
k_implsch4(const DevTab<T>* __restrict__ tp, int kijs, int kijl, T* __restrict__ fl1, const T* __restrict__ wvprpt, T* __restrict__ ffa,
           T* __restrict__ intfa, int* __restrict__ mij_out, T* __restrict__ xllws, T* __restrict__ fin, T* __restrict__ gfast, int gk,
           T* __restrict__ wi, const V4Adv<T> adv) {
  // scalar slots of the point's LDS row that are free in PART 2 (STRESSO's temporaries) carry the hand-over values
  enum { C2_FMEAN = C_XSN, C2_FMEANWS = C_YSN, C2_AKMEAN = C_UST, C2_XKMEAN = C_SINU, C2_SC = C_COSU };
  // gfast (optional): compact rows gfast[ij][K][gk] that also receive the first gk frequencies of the new spectrum -- what the next
  // advection step's fast-wave sub-steps start from (ecwam_hip_set_fastwave_copy): written from the tile, no second pass over FL1
  constexpr int G = NANG / 2, NFRE = V4_NFRE, N = NANG * NFRE, RS = PP * NANG, NS = V4_NS(NANG);
  constexpr int NSH = (NH + 1) / 2;          // even shifts -2 NSH .. 2 NSH cover the taps -NH .. NH+1 and the DIA rotations
  constexpr int NTAP = 2 * NH + 1;
  constexpr int VEC = 16 / (int)sizeof(T);   // elements per 16-byte global access
  constexpr int NC = NFRE / VEC;             // 16-byte chunks per direction
  static_assert(PP * G <= 64 && NFRE % VEC == 0 && R2 + 2 <= 2 * NSH + 1, "layout");
  constexpr int PLN = V4_PLN(PP, NANG);      // elements of a plane [PP][NFRE] that doubles as a staging row [PP][NANG]
  constexpr bool EVEN = (NFRE % G == 0);     // every lane owns NS frequencies (not at 48 directions)
  // frequency q of the lane, and whether the lane really owns it (48 directions: q = 1 exists on the lanes j < 12 only; the others repeat
  // frequency NFRE -- table fills then store the same values twice, sums take a weight of zero)
  auto mq = [&](int q, int jj) -> int { const int m = q * G + jj; return (EVEN || m < NFRE) ? m : NFRE - 1; };
  auto okq = [&](int q, int jj) -> bool { return EVEN || q * G + jj < NFRE; };
  static_assert(RARE || !(EXT && ENHMC), "ISNONLIN = 1 beside LLGCBZ0 / LLNORMAGAM: the RARE build");
  typedef T VT __attribute__((ext_vector_type(VEC)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const DevTab<T>& tb = *tp;
  T* sT = reinterpret_cast<T*>(smem_raw);          // [NFRE + V4_NSTG][PP][NANG]
  T* sStg = sT + NFRE * RS;                        // staging rows 0..3
  T* sFac4 = sT + (NFRE + V4_NSTG) * RS;           // [PP][NFRE][4]
  T* sPl = sFac4 + PP * NFRE * 4;                  // [2][PLN]: SQRT(WAVNUM), LOG(WAVNUM Z0M) as [PP][NFRE]; staging rows 4, 5 during the sweep
  T* sSC = sPl + 2 * PLN;                          // [PP][NSC]
  V4Ctx<T, NANG, PP> L;
  L.lane = threadIdx.x & 63;
  const int lane = L.lane;
  {
    // spare lanes (64 - PP G of them) shadow a lane of their own half-wave (a 64-bit LDS access is served in two groups of 32
    // lanes; equal addresses inside a group are one broadcast access): no bank conflicts from the shadows
    if constexpr (G == 18) {
      static_assert(G != 18 || PP == 3, "36 directions: one point per DPP row + two extra lanes per point");
      if (lane < 48) { L.p = lane >> 4; L.j = lane & 15; }
      else if (lane < 54) { L.p = (lane - 48) >> 1; L.j = 16 + ((lane - 48) & 1); }
      else { L.p = 2; L.j = 0; }                       // shadows of lane 32
    } else {
      int src = lane;
      if (lane >= PP * G) {
        if (PP * G > 32) src = lane >= 32 ? 32 : 0;
        else { src = lane >= 32 ? lane - 32 : 0; if (src >= PP * G) src = 0; }
      }
      L.p = src / G;
      L.j = src - L.p * G;
    }
  }
  const int p = L.p, j = L.j;
  // ADV: the workgroups of a launch are dealt round-robin to the 8 XCDs; XCD x takes the contiguous eighth [x gridDim.x / 8, ...) of the
  // wave's triples so that a row fetched as somebody's neighbour is met again in the same L2 (gridDim.x is a multiple of 8)
  // (xcd_walk = G > 1: groups of G consecutive waves per XCD, the groups of the 8 XCDs interleaved -- the chip still moves through the grid as
  // one front; gridDim.x is then a multiple of 8 G)
  int blk = (int)blockIdx.x;
  if (ADV != 0 && adv.xcd_walk == 1) blk = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  else if (ADV != 0 && adv.xcd_walk > 1) {
    const int G_ = adv.xcd_walk, x_ = (int)(blockIdx.x & 7u), j_ = (int)(blockIdx.x >> 3);
    blk = (j_ / G_) * (8 * G_) + x_ * G_ + (j_ % G_);
  }
  const int ij0 = kijs + blk * PP;
  if (ij0 >= kijl) return;
  const int n = kijl - ij0 < PP ? kijl - ij0 : PP;   // points of this wave; a short last wave replicates its last point
  const int ij = ij0 + (p < n ? p : n - 1);
  // Position of direction k of point q inside a row of PP NANG elements: q NANG + k.  (A split layout -- the first 32 directions of the three
  // points at 0, 32, 64, their last four at 96, 100, 104, so that a half-wave's own pairs land on 64 different banks -- measured MORE
  // conflict cycles and 1 % more time in round 5: profiles/r05_lds_row_layout.txt.)
  auto rowpos = [&](int q, int k) -> int { return q * NANG + k; };
  L.tile = sT; L.own = rowpos(p, 2 * j);
  L.fac4 = sFac4 + p * NFRE * 4; L.sq = sPl + p * NFRE; L.zcn = sPl + PLN + p * NFRE; L.c = sSC + p * NSC;
  int sh[2 * NSH + 1];
#pragma unroll
  for (int i = 0; i <= 2 * NSH; i++) {
    int k = 2 * j + 2 * (i - NSH);
    k = k < 0 ? k + NANG : (k >= NANG ? k - NANG : k);
    sh[i] = rowpos(p, k);
  }
  if constexpr (G == 18) {
    const bool low = lane < 48 && j < 2;
    L.rot.a0 = 4 * (low ? 48 + 2 * p + j : lane);
    L.rot.a1 = 4 * (lane < 48 ? lane : 16 * p);
    L.rot.a2 = L.rot.a3 = L.rot.a4 = 0;
    L.rot.fold = low ? T(1) : T(0);
  } else {
    const int base = p * G;
#define V4_ROT(r) (4 * (base + ((j + (r)) >= G ? j + (r) - G : j + (r))))
    if (G == 24) { L.rot.a0 = V4_ROT(12); L.rot.a1 = V4_ROT(6); L.rot.a2 = V4_ROT(3); }
    if (G == 12) { L.rot.a0 = V4_ROT(6); L.rot.a1 = V4_ROT(3); L.rot.a2 = 0; }
    if (G == 6) { L.rot.a0 = V4_ROT(3); L.rot.a1 = 0; L.rot.a2 = 0; }
    L.rot.a3 = V4_ROT(1); L.rot.a4 = V4_ROT(2);
    L.rot.fold = T(0);
#undef V4_ROT
  }
  {
    const int mi = lane < NFRE ? lane : 0;
    L.rDFIM = tb.DFIM[mi]; L.rDFIMOFR = tb.DFIMOFR[mi]; L.rDFIMFR = tb.DFIMFR[mi]; L.rZPIFR = tb.ZPIFR[mi]; L.rRHOWG = tb.RHOWG_DFIM[mi];
    L.rCOFRM4 = tb.COFRM4[mi]; L.rFLMAX = tb.FLMAX[mi];
  }
  // V4_RECPF: the module constants of a row from the lane-held copies of the tables above (v_readlane) instead of a scalar load per row
  // that is waited for where it is issued (the row loops of SDEPTHLIM / FKMEAN and FEMEANWS)
  constexpr bool RLANE = (V4_RECPF != 0) && sizeof(T) == 4;
  L.sinth = V2<T>{tb.SINTH[2 * j], tb.SINTH[2 * j + 1]};
  L.costh = V2<T>{tb.COSTH[2 * j], tb.COSTH[2 * j + 1]};
  T* c = L.c;
  const T* tF = sT + L.own;        // the lane's pair in row 0
  T* tFw = sT + L.own;

  // ---- spectra F[ij][K][M] -> tile [M][point][K]: 16-byte global loads (chunk w of a point = direction w / NC, frequencies
  //      VEC (w % NC) ..), the VEC frequencies of a chunk go to VEC rows; all the loads are issued before the first LDS store
  //      -- and before the loads of the point scalars and factor tables below, whose LDS stores come first: one round trip to memory
  //      for all of them instead of one after the other
  constexpr int NVL = N / VEC, NITL = (NVL + 63) / 64;   // chunks per point, iterations per point
  VT val[ADV != 0 ? 1 : PP][ADV != 0 ? 1 : NITL];
  if constexpr (ADV != 0) {
    // the tile is the advected spectrum: staging rows, factor table and planes are free until the tables below are filled
    static_assert(PART == 0, "the advecting tile load belongs to the one-kernel build");
    v4_advect_tile<T, NANG, PP, ADV, V4_NSTG * RS + PP * NFRE * 4 + 2 * PLN>(adv, ij0, n, lane, sT, sStg);
  } else {
#pragma unroll
    for (int q = 0; q < PP; q++) {
      const T* g = fl1 + (size_t)(ij0 + (q < n ? q : n - 1)) * N;
#pragma unroll
      for (int it = 0; it < NITL; it++) {
        const int w = lane + 64 * it;
        if (w < NVL) val[q][it] = *reinterpret_cast<const VT*>(g + (size_t)w * VEC);
      }
    }
  }
  // ---- point scalars (sinflx.F90:105-122; the first TAUT_Z0 ran in k_implsch4_pre), one lane per point: the loads are unconditional
  //      (every lane, of point lane mod PP) so that they are in flight with the tile's; the same for the factor tables below
  const int pl = lane < PP ? lane : lane % PP;
  const int pid = ij0 + (pl < n ? pl : n - 1);
  V4_CHK(n >= 1 && n <= PP && p >= 0 && p < PP && j >= 0 && j < G && ij >= kijs && ij < kijl && pid >= kijs && pid < kijl);
  const T* ffp = ffa + (size_t)pid * ECWAM_HIP_NFF;
  const T* frp = fin + (size_t)pid * V4_NFIN;
  const T p_aird = ffp[0], p_wdwave = ffp[1], p_ci = ffp[2], p_wswave = ffp[3], p_wstar = ffp[4], p_tauw = ffp[8], p_tauwdir = ffp[9];
  const T p_emaxdpt = ffp[14], p_depth = ffp[15];
  const T p_sinwd = frp[FIN_SINWD], p_coswd = frp[FIN_COSWD], p_rnfac = frp[FIN_RNFAC], p_cosdiff = frp[FIN_COSDIFF];
  const T p_ufric = frp[FIN_UFRIC], p_z0m = frp[FIN_Z0M], p_z0b = frp[FIN_Z0B], p_chrnck = frp[FIN_CHRNCK];
  T h_o[PART == 2 ? 9 : 1];   // PART 2: what PART 1 handed over (loaded with the rest, unconditionally)
  if constexpr (PART == 2) {
    h_o[0] = frp[FIN_MIJ]; h_o[1] = frp[FIN_SDS]; h_o[2] = frp[FIN_EMEAN]; h_o[3] = frp[FIN_F1MEAN]; h_o[4] = frp[FIN_FMEAN];
    h_o[5] = frp[FIN_FMEANWS]; h_o[6] = frp[FIN_AKMEAN]; h_o[7] = frp[FIN_XKMEAN]; h_o[8] = frp[FIN_SC];
  }
  const T* wp = wvprpt + (size_t)ij * ECWAM_HIP_NWPR * NFRE;
  T w_wn[NS], w_cg[NS], w_ci[NS], w_xk[NS];   // WAVNUM, CGROUP, CINV, XK2CG of the lane's frequencies m = s G + j
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const int m = mq(q, j);
    w_wn[q] = wp[m]; w_cg[q] = wp[NFRE + m]; w_ci[q] = wp[2 * NFRE + m]; w_xk[q] = wp[3 * NFRE + m];
  }
  const T DEPTHv = ffa[(size_t)ij * ECWAM_HIP_NFF + 15];
  const T CICOVERi = ffa[(size_t)ij * ECWAM_HIP_NFF + 2], CITHICKi = ffa[(size_t)ij * ECWAM_HIP_NFF + 13];
  if (lane < PP) {
    T* q = sSC + lane * NSC;
    // friction-velocity forcing (ICODE = 1, 2): the 10 m wind is the log-profile wind k_implsch4_pre derived (airsea.F90:107-115)
    q[C_AIRD] = p_aird; q[C_WDWAVE] = p_wdwave; q[C_WSWAVE] = (RARE && tb.ICODE != 3) ? frp[FIN_WSWAVE] : p_wswave; q[C_WSTAR] = p_wstar;
    q[C_TAUW] = p_tauw; q[C_TAUWDIR] = p_tauwdir;
    q[C_RAORW] = m_max(p_aird, T(1)) * tb.ROWATERM1; q[C_EMAXDPT] = p_emaxdpt; q[C_DEPTH] = p_depth;
    q[C_SINWD] = p_sinwd; q[C_COSWD] = p_coswd; q[C_RNFAC] = p_rnfac;
    if (EXT && tb.LLGCBZ0) q[C_TWCOS] = p_cosdiff;
    q[C_UFRIC] = p_ufric; q[C_Z0M] = p_z0m; q[C_Z0B] = p_z0b; q[C_CHRNCK] = p_chrnck;
    q[C_SPARE] = p_ci;   // CICOVER
    if constexpr (PART == 2) {
      q[C_MIJ] = h_o[0]; q[C_SDS] = h_o[1]; q[C_EMEAN] = h_o[2]; q[C_F1MEAN] = h_o[3];
      q[C2_FMEAN] = h_o[4]; q[C2_FMEANWS] = h_o[5]; q[C2_AKMEAN] = h_o[6]; q[C2_XKMEAN] = h_o[7]; q[C2_SC] = h_o[8];
    }
  }
  // RARE (ice radiative stress): FLDICE(M) = -ALP(M) CGROUP(M) of the last of SDICE1 / SDICE3 that is active, without the ice cover factor
  // (SLICE of sdice1.F90:177 / sdice3.F90:143), for the lane's frequencies M = q G + j + 1; handed out row by row with one lane exchange.
  // (A vector value, not an array: an array read through a chain of selects ends as an indexed access to scratch memory.)
  typedef T FLIV __attribute__((ext_vector_type(RARE ? NS : 1)));
  FLIV rFLI = T(0);
  // ---- per-frequency factors of the point: lane j fills M = j+1, j+1+G, ...
  {
    T CITHICKv = T(0), DINV = T(0), ALPFAC = tb.ZALPFACX;      // ice thickness, inverse floe diameter, broken-ice factor of the point
    // sea-ice attenuation (implsch.F90:312-339): SDICE1 (scattering, sdice1.F90:104-181) and SDICE3 (viscous friction, sdice3.F90:110-160)
    // are, like SBOTTOM, a damping rate per (point, frequency): FLD += c, SL += c F.  The three rates share one slot of the table (the
    // reference adds them one after the other: the sums differ in rounding only).  SDICE2 depends on F itself and the NEMO coupling
    // needs SLICE on its own: the RARE build.
    const bool ice1 = tb.LICERUN && tb.LCIWA1, ice3 = tb.LICERUN && tb.LCIWA3;
    T CICOVERv = T(0);
    if (ice1 || ice3) {
      CICOVERv = CICOVERi;
      CITHICKv = CITHICKi;
      // broken ice attenuates less (icebreak_modify_attenuation.F90:82-93); IBRMEM is the input slot 15 of INTFLDS
      if (tb.LWNEMOCOUIBR && intfa[(size_t)ij * ECWAM_HIP_NINTF + 15] <= tb.ZIBRW_THRSH) ALPFAC = T(1) / tb.ZALPFACX;
    }
    if (ice1) {   // mean floe diameter of the fragmentation cascade
      const T CIFRGL = T(0.955), CIDMIN = T(20.0), CIFRGMT = T(2.0), A = T(200.0), C = T(300.0);
      const int MAXICM = (int)(m_log(A / CIDMIN) / m_log(CIFRGMT));
      DINV = CIDMIN;
      if (CITHICKv > T(0)) {
        const T CIDMAX = A + C * CICOVERv;
        int ICM = (int)(m_log(CIDMAX / CIDMIN) / m_log(CIFRGMT));
        if (ICM > MAXICM) ICM = MAXICM;
        T SN = T(0), SD = T(0), X = T(1), FI = T(1);
        for (int I = 0; I <= ICM; I++) {   // X = (CIFRGMT**2*CIFRGL)**I, FI = CIFRGMT**I
          SN = SN + X * CIDMAX / FI;
          SD = SD + X;
          X = X * (CIFRGMT * CIFRGMT * CIFRGL);
          FI = FI * CIFRGMT;
        }
        DINV = T(1) / (SN / SD);
      }
    }
#pragma unroll
    for (int qq = 0; qq < NS; qq++) {
      const int m = mq(qq, j);
      T* f = L.fac4 + m * 4;
      const T WAVNUM = w_wn[qq], XK2CG = w_xk[qq];
      f[Q4_WAVNUM] = WAVNUM; f[Q4_CINV] = w_ci[qq]; f[Q4_BSC] = WAVNUM * (T(1) / tb.ZPI) * XK2CG; L.sq[m] = m_sqrt(WAVNUM);
      T sbo = T(0);   // sbottom.F90:79-89
      if (m < tb.NFRE_RED && DEPTHv < tb.BATHYMAX) sbo = (-T(2) * T(0.038) * tb.GM1) * WAVNUM / m_sinh(m_min(T(2) * DEPTHv * WAVNUM, T(50)));
      if (ice1 || ice3) {
        const T CGROUP = w_cg[qq];
        T dmp = T(0);
        if (ice1 && CITHICKv > T(0)) {
          const T d1 = -v4_sdice1_alp(tb, m, CITHICKv, DINV) * CGROUP;
          dmp = CICOVERv * d1;
          if constexpr (RARE) rFLI[qq] = d1;
        }
        if (ice3) {
          const T ALP = v4_sdice3_alp(tb, m, CITHICKv, ALPFAC);
          dmp = dmp + (-CICOVERv * ALP * CGROUP);
          if constexpr (RARE) rFLI[qq] = -ALP * CGROUP;
        }
        sbo = dmp + sbo;
      }
      f[Q4_SBO] = sbo;
    }
  }
  if constexpr (ADV == 0) {   // the tile: registers -> LDS
    int k = lane / NC, r = lane - k * NC;
#pragma unroll
    for (int it = 0; it < NITL; it++) {
      const int w = lane + 64 * it;
      if (w < NVL) {
        T* d = sT + (r * VEC) * RS + rowpos(0, k);
        const int qo = NANG;      // distance of the next point's direction k
#pragma unroll
        for (int q = 0; q < PP; q++)
#pragma unroll
          for (int i = 0; i < VEC; i++) d[i * RS + q * qo] = val[q][it][i];
      }
      k += 64 / NC; r += 64 % NC;
      if (r >= NC) { r -= NC; k += 1; }
    }
  }
  WSYNC();
  V4_PHASE_EXIT(201);
  const T AIRD = c[C_AIRD], WSWAVE = c[C_WSWAVE], RAORW = c[C_RAORW], EMAXDPT = c[C_EMAXDPT], DEPTH = c[C_DEPTH];
  const T sinwd = c[C_SINWD], coswd = c[C_COSWD], CICOVER = c[C_SPARE];
  const V2<T> coswdif = L.costh * coswd + L.sinth * sinwd;
  const T frl = tb.FR[NFRE - 1];
  const T DELT25 = tb.WETAIL * frl * tb.DELTH;
  const V2<T> z2 = {T(0), T(0)};
  const V2<T> cpos = {m_max(T(0), coswdif.x), m_max(T(0), coswdif.y)};
  const V2<T> FLM = ((T(1) - T(0.9) * m_min(CICOVER, T(0.99))) * tb.FLMIN) * (cpos * cpos);

  // FKMEAN (fkmean.F90:94-150) from per-lane sums over M and one all-reduce per pair of quantities
  auto fkmean_finish = [&](V2<T> s0, V2<T> s1, V2<T> s2, T& EM, T& FM1, T& F1, T& AK, T& XK) {
    s0 = v4_allsum<G, T>(s0, L.rot); s1 = v4_allsum<G, T>(s1, L.rot); s2 = v4_allsum<G, T>(s2, L.rot);
    const T COEFM1 = tb.FRTAIL * tb.DELTH;
    const T COEF1 = tb.WP1TAIL * tb.DELTH * frl * frl;
    const T COEFA = COEFM1 * m_sqrt(tb.G) / tb.ZPI;
    const T COEFX = COEF1 * (tb.ZPI / m_sqrt(tb.G));
    const T tl = s2.y;
    EM = tb.EPSMIN + s0.x + DELT25 * tl;
    FM1 = EM / (tb.EPSMIN + s0.y + COEFM1 * tl);
    F1 = (tb.EPSMIN + s1.x + COEF1 * tl) / EM;
    AK = tb.EPSMIN + s1.y + COEFA * tl;
    AK = (EM / AK) * (EM / AK);
    XK = tb.EPSMIN + s2.x + COEFX * tl;
    XK = (XK / EM) * (XK / EM);
  };

  // ---- SDEPTHLIM (sdepthlim.F90:64-78, semean.F90:82-120), FKMEAN, the tail of sinflx.F90:124-128 and the orbital integrals of the
  //      swell damping (sinput_ard.F90:213-222) in two passes over the tile
  T EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN;
  T SCL = T(1);   // SDEPTHLIM's scale of the point (handed to PART 2)
  if constexpr (PART == 2) {
    // the spectrum as PART 1 left it in its tile: scaled, floored, the last row raised to the noise floor (the same operations)
    EMEAN = c[C_EMEAN]; F1MEAN = c[C_F1MEAN]; FMEAN = c[C2_FMEAN]; AKMEAN = c[C2_AKMEAN]; XKMEAN = c[C2_XKMEAN];
    const T sc = c[C2_SC];
    const T flo = tb.LBIWBK ? tb.EPSMIN : -std::numeric_limits<T>::infinity();
    for (int m = 0; m < NFRE; m++) {
      V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS) * sc;
      f.x = m_max(f.x, flo); f.y = m_max(f.y, flo);
      if (m == NFRE - 1) { f.x = m_max(f.x, FLM.x); f.y = m_max(f.y, FLM.y); }
      *reinterpret_cast<V2<T>*>(tFw + m * RS) = f;
    }
  } else {
    T sc = T(1);
    if (tb.LBIWBK) {
      V2<T> s = z2;
      for (int m = 0; m < NFRE; m++) {
        const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
        s.x = s.x + (RLANE ? lane_get(L.rDFIM, m) : tb.SINROW[m][1]) * (f.x + f.y);
      }
      {
        const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + (NFRE - 1) * RS);
        s.y = f.x + f.y;
      }
      s = v4_allsum<G, T>(s, L.rot);
      const T EM = tb.EPSMIN + s.x + DELT25 * s.y;
      sc = m_min(EMAXDPT / EM, T(1));
    }
    SCL = sc;
    V2<T> s0 = z2, s1 = z2, s2 = z2, so = z2;
    // branch-free rows: without LBIWBK the scale is 1 and the floor -infinity (both exact), the row is stored back unchanged; the
    // last row (raised to the noise floor for the orbital integrals only) is peeled off
    const T flo = tb.LBIWBK ? tb.EPSMIN : -std::numeric_limits<T>::infinity();
    auto row202 = [&](int m, bool last) {
      V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS) * sc;
      f.x = m_max(f.x, flo); f.y = m_max(f.y, flo);
      const T t = f.x + f.y;
      const T* row = tb.SINROW[m];
      const T dfm = RLANE ? lane_get(L.rDFIM, m) : row[1], sqm = L.sq[m], sig = RLANE ? lane_get(L.rZPIFR, m) : row[0];
      s0 = s0 + V2<T>{dfm, RLANE ? lane_get(L.rDFIMOFR, m) : row[5]} * t;
      s1 = s1 + V2<T>{RLANE ? lane_get(L.rDFIMFR, m) : row[6], fs_div<32>(dfm, sqm)} * t;
      s2.x = s2.x + (sqm * dfm) * t;
      if (last) {
        s2.y = t;
        f.x = m_max(f.x, FLM.x); f.y = m_max(f.y, FLM.y);   // the orbital integrals see the raised tail
      }
      const T to = f.x + f.y;
      so = so + V2<T>{dfm * (sig * sig), dfm} * to;
      *reinterpret_cast<V2<T>*>(tFw + m * RS) = f;
    };
    for (int m = 0; m < NFRE - 1; m++) row202(m, false);
    row202(NFRE - 1, true);
    fkmean_finish(s0, s1, s2, EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN);
    so = v4_allsum<G, T>(so, L.rot);
    if (j == 0) { c[C_UORBT] = tb.EPSMIN + so.x; c[C_AORB] = tb.EPSMIN + so.y; c[C_EMEAN] = EMEAN; c[C_F1MEAN] = F1MEAN; }
  }
  if constexpr (JAN) {
    // SDISSIP_JAN (sdissip_jan.F90:92-128) is a rate per (point, frequency): TEMP1(M) goes to the slot of the saturation scale, which
    // nothing reads before IMPHFTAIL (restored behind the sweep)
    const T SDSJ = (tb.CDIS * tb.ZPI) * F1MEAN * (EMEAN * EMEAN) * m_pow4(XKMEAN);
    const T CVIS = tb.RNU * tb.CDISVIS;
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const int m = mq(q, j);
      const T wn = L.fac4[m * 4 + Q4_WAVNUM];
      const T X = wn / XKMEAN;
      L.fac4[m * 4 + Q4_BSC] = SDSJ * X * ((T(1) - tb.DELTA_SDIS) + tb.DELTA_SDIS * X) + CVIS * (wn * wn);
    }
  }
  WSYNC();
  V4_PHASE_EXIT(202);
  const bool gcb = EXT && tb.LLGCBZ0, norma = EXT && tb.LLNORMAGAM;
  V2<T> sinwdif2 = z2;
  if (norma) {
    const V2<T> sd = L.sinth * coswd - L.costh * sinwd;   // SIN(TH - WDWAVE)
    sinwdif2 = sd * sd;
  }
  // gravity-capillary TAUT_Z0 of the points of the wave, four at a time: a DPP row of 16 lanes per point (rows beyond the last point
  // shadow it: same values, same stores)
  auto taut_z0_gc = [&](int iusfg) {
    for (int b = 0; b < (PP + 3) / 4; b++) {
      const int q4 = 4 * b + (lane >> 4);
      T* cq = sSC + (q4 < PP ? q4 : PP - 1) * NSC;
      T UF = cq[C_UFRIC], Z0 = cq[C_Z0M], Z0Bv = cq[C_Z0B], CH = cq[C_CHRNCK];
      const T cosd = iusfg ? cq[C_COSWD] * cq[C_TWCOS] + cq[C_SINWD] * cq[C_TWSIN] : cq[C_TWCOS];
      const T WSW = cq[C_WSWAVE], WST = cq[C_WSTAR];
      taut_z0_b_rows(tb, lane & 15, iusfg, cq[C_HALP], WSW, cosd, cq[C_TAUW], cq[C_RNFAC], UF, Z0, Z0Bv, CH);
      T sgn = T(0);
      if (iusfg) sgn = wsigstar(tb, WSW, UF, Z0, WST);
      WSYNC();
      if ((lane & 15) == 0) {
        cq[C_UFRIC] = UF; cq[C_Z0M] = Z0; cq[C_Z0B] = Z0Bv; cq[C_CHRNCK] = CH;
        if (iusfg) cq[C_SIGN] = sgn;
      }
      WSYNC();
    }
  };
  if (PART != 2 && gcb) {
    // HALPHAP (halphap.F90:68-112, meansqs_lf.F90:80-100, femean.F90:84-121): Phillips parameter of the wind-sea half plane
    const V2<T> wd = {__builtin_signbit(coswdif.x) ? T(0) : T(1), __builtin_signbit(coswdif.y) ? T(0) : T(1)};
    V2<T> sa = z2, sb = z2;   // (XMSS, EM), (FM, last row of MAX(F WD, EPSMIN))
    T f1d = T(0);
    for (int m = 0; m < NFRE; m++) {
      const V2<T> v = *reinterpret_cast<const V2<T>*>(tF + m * RS) * wd;
      const T t1 = v.x + v.y, t2 = m_max(v.x, tb.EPSMIN) + m_max(v.y, tb.EPSMIN);
      const T dfm = lane_get(L.rDFIM, m), wn = L.fac4[m * 4 + Q4_WAVNUM];
      sa = sa + V2<T>{(dfm * wn * wn) * t1, dfm * t2};
      sb.x = sb.x + lane_get(L.rDFIMOFR, m) * t2;
      if (m == NFRE - 1) { sb.y = t2; f1d = t1; }
    }
    sa = v4_allsum<G, T>(sa, L.rot); sb = v4_allsum<G, T>(sb, L.rot);
    const T F1D = tb.DELTH * v4_allsum1<G, T>(f1d, L.rot);
    const T XMSS = sa.x;
    const T EM = sa.y + tb.WETAIL * frl * tb.DELTH * sb.y;
    T FM = sb.x + tb.FRTAIL * tb.DELTH * sb.y;
    FM = m_max(EM / FM, tb.FR[0]);
    const T ATAIL = tb.ZPI4GM2 * tb.FR5[NFRE - 1] * F1D;
    T ALPHAP;
    if (EM > T(0) && FM < tb.FR[NFRE - 3]) {
      ALPHAP = XMSS / (m_log(tb.FR[NFRE - 1]) - m_log(FM));
      if (ALPHAP > tb.ALPHAPMAX) ALPHAP = ATAIL;
    } else ALPHAP = ATAIL;
    if (j == 0) c[C_HALP] = T(0.5) * m_min(ALPHAP, tb.ALPHAPMAX);
    WSYNC();
    if (!(RARE && tb.ICODE != 3)) taut_z0_gc(0);     // (friction-velocity forcing: Z0WAVE in k_implsch4_pre took the place of the first TAUT_Z0)
  }
  T UFRIC = c[C_UFRIC], Z0M = c[C_Z0M];
  T RNFAC = c[C_RNFAC];

  auto femws_finish = [&](V2<T> wse, V2<T> wslast, T& FM, T& EMW) {
    const V2<T> s = v4_allsum<G, T>(wse, L.rot);
    const T t2 = v4_allsum1<G, T>(wslast.x + wslast.y, L.rot);
    const T em = tb.EPSMIN + s.x + DELT25 * t2;
    const T fm = tb.EPSMIN + s.y + (tb.FRTAIL * tb.DELTH) * t2;
    FM = em / fm;
    EMW = em;
  };
  auto frcutindex4 = [&](T FMEANWS, T UF) -> int {   // frcutindex.F90:84-97
    const T FPMH = tb.TAILFACTOR / tb.FR[0];
    const T FPPM = tb.TAILFACTOR_PM * tb.G / (tb.FRIC * tb.ZPIFR[0]);
    int MIJ = NFRE;
    if (CICOVER <= tb.CITHRSH_TAIL) {
      const T FPM4 = m_max(m_max(FMEANWS, FMEAN) * FPMH, FPPM / m_max(UF, tb.EPSMIN));
      MIJ = m_nint(m_log10(FPM4) * tb.FLOGSPRDM1) + 1;
      MIJ = MIJ < 1 ? 1 : (MIJ > NFRE ? NFRE : MIJ);
    }
    return MIJ;
  };
  auto rrh = [&](int m, int MIJ) -> T {   // RHOWGDFTH(M), zero above MIJ, halved at MIJ (frcutindex.F90:98-107)
    // RHOWG_DFIM(M) from the lane table (lane m holds it): an exchange through the LDS crossbar instead of a round trip to the table in memory
    T r = v4_bp(4 * m, L.rRHOWG);
    if (m + 1 == MIJ && MIJ != NFRE) r = T(0.5) * r;
    return (m + 1 <= MIJ) ? r : T(0);
  };
  T rX[NS], rY[NS];
  T* sXY = sStg + p * NFRE * 2;   // 36 directions: [M][2] row integrals X, Y of the point (the staging rows are idle outside the sweep)
  // row table [M] of the point for the positive input of SINPUT's second call (staging rows 2, 3 are idle outside the sweep); lanes that
  // hold no row total of their own (the extras of the 36-direction layout and their shadows) store to a private slot behind it
  constexpr int SPOFF = (G == 18) ? 2 * RS : 0;   // (36 directions: rows 0, 1 hold the row integrals X, Y of the stress)
  static_assert(SPOFF + PP * V4_NFRE <= (G == 18 ? 3 : 4) * RS, "the row tables fit the staging rows");
  T* gsp = (G == 18 && lane >= 48) ? sStg + 3 * RS + lane : sStg + SPOFF + p * NFRE;
  // stress sums below the cut-off and the F(:,MIJ) integrals of TAU_PHI_HF (stresso.F90:148-173, tau_phi_hf.F90:170-196)
  auto post_stress = [&](int MIJ, V2<T> apl, bool phiwa) {
    V4_CHK(MIJ >= 1 && MIJ <= NFRE);
    const T zpm = tb.ZPIFR[MIJ - 1], f5m = tb.FR5[MIJ - 1];   // for STRESSO: in flight behind the sums below
    V2<T> s = z2;
    T sp = T(0);
    const T* spt = sStg + SPOFF + p * NFRE;
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const int m = mq(q, j);
      const T w = okq(q, j) ? rrh(m, MIJ) : T(0);
      const T wx = w * L.fac4[m * 4 + Q4_CINV];
      if constexpr (G == 18) s = s + wx * *reinterpret_cast<const V2<T>*>(sXY + 2 * m);
      else s = s + V2<T>{wx * rX[q], wx * rY[q]};
      if (phiwa) sp += w * spt[m];   // the lane weighs the row totals of the positive input it owns (m = q G + j): the all-reduce below adds the lanes up
    }
    s = v4_allsum<G, T>(s, L.rot);
    T PH = T(0);
    if (phiwa) PH = v4_allsum1<G, T>(apl.x + apl.y + sp, L.rot);
    const V2<T> fm = *reinterpret_cast<const V2<T>*>(tF + (MIJ - 1) * RS);
    const V2<T> fc2 = fm * cpos * cpos, fc3 = fc2 * cpos;
    const V2<T> h = v4_allsum<G, T>(V2<T>{fc3.x + fc3.y, fc2.x + fc2.y}, L.rot);
    V2<T> hn = z2;   // F1DSIN2, F1D of the normalised growth rate (tau_phi_hf.F90:187-196)
    if (norma) {
      const V2<T> fs = fm * sinwdif2;
      hn = v4_allsum<G, T>(V2<T>{fs.x + fs.y, fm.x + fm.y}, L.rot);
    }
    if (j == 0) {
      c[C_XS] = s.x; c[C_YS] = s.y; c[C_F1DCOS3] = tb.DELTH * h.x; c[C_F1DCOS2] = tb.DELTH * h.y; c[C_F1DSIN2] = tb.DELTH * hn.x; c[C_F1D] = tb.DELTH * hn.y;
      c[C_MIJ] = (T)MIJ; c[C_ZPIFRMIJ] = zpm; c[C_FR5MIJ] = f5m;
      if (phiwa) c[C_PHIWA] = PH;
    }
  };

  // ---- first SINFLX call (sinflx.F90:105-183): MIJ and the wave stress only
  unsigned long long xm0 = 0ull, xm1 = 0ull;
  V2<T> wse, wslast, apl;
  T FMEANWS, EMW;
  int MIJ;
  T SDS;
  T* gx = (PART == 0 ? xllws : wi) + (size_t)ij * N + 2 * j;   // this lane's pair in row 0 of the block that holds the wind-input coefficient
  if constexpr (PART != 2) {
  if constexpr (JAN) v4_sinput_jan<T, NANG, PP, 1, false, EXT>(tb, L, UFRIC, Z0M, RAORW, T(0), coswdif, nullptr, nullptr, xm0, xm1, wse, wslast, apl, rX, rY, sXY,
                                                                 norma, sinwdif2, wp + 3 * NFRE, RNFAC);
  else if (norma) v4_sinput_n<T, NANG, PP, 1, false>(tb, L, wp + 3 * NFRE, UFRIC, Z0M, RAORW, RNFAC, T(0), T(0), T(0), T(0), coswdif, sinwdif2, nullptr, nullptr,
                                                    xm0, xm1, wse, wslast, apl, rX, rY, sXY);
  else v4_sinput<T, NANG, PP, 1, false>(tb, L, UFRIC, Z0M, RAORW, T(0), T(0), T(0), T(0), sinwd, coswd, nullptr, nullptr, xm0, xm1, wse, wslast, apl, rX, rY, sXY);
  femws_finish(wse, wslast, FMEANWS, EMW);
  MIJ = frcutindex4(FMEANWS, UFRIC);
  post_stress(MIJ, apl, false);
  WSYNC();
  V4_PHASE_EXIT(203);
  // ---- stage 2: STRESSO scalars, second TAUT_Z0, WSIGSTAR, swell set-up, SDIWBK
  v4_stresso<T, PP, EXT>(tb, sSC, lane, false);
  WSYNC();
  if constexpr (RARE) {
    if (tb.ICODE != 3 && norma && tb.LLCAPCHNK) {   // second SINFLX call: RNFAC from the log-profile wind (sinflx.F90:116-120)
      if (lane < PP) {
        T* q = sSC + lane * NSC;
        q[C_RNFAC] = T(1) + tb.DTHRN_A * (T(1) + m_tanh(q[C_WSWAVE] - tb.DTHRN_U));
      }
      WSYNC();
      RNFAC = c[C_RNFAC];
    }
  }
  if (gcb) taut_z0_gc(1);   // (with WSIGSTAR)
  if (lane < PP) {
    T* q = sSC + lane * NSC;
    if (!gcb) {
      T UF = q[C_UFRIC], Z0 = q[C_Z0M], Z0Bv = q[C_Z0B], CH = q[C_CHRNCK];
      taut_z0_c(tb, 1, q[C_WSWAVE], q[C_COSWD] * q[C_TWCOS] + q[C_SINWD] * q[C_TWSIN], q[C_TAUW], UF, Z0, Z0Bv, CH);
      q[C_UFRIC] = UF; q[C_Z0M] = Z0; q[C_Z0B] = Z0Bv; q[C_CHRNCK] = CH;
      q[C_SIGN] = wsigstar(tb, q[C_WSWAVE], UF, Z0, q[C_WSTAR]);
    }
    swell_setup_pt(tb, q);
    q[C_SDS] = sdiwbk_pt(tb, q[C_EMAXDPT], q[C_EMEAN], q[C_F1MEAN], q[C_DEPTH]);
  }
  WSYNC();
  UFRIC = c[C_UFRIC]; Z0M = c[C_Z0M];
  SDS = c[C_SDS];
  V4_PHASE_EXIT(204);
  // ---- second SINFLX call: wind-input coefficient (parked in the point's XLLWS block, [M][K]; PART 1: in the rows of wi), XLLWS, MIJ,
  //      wave stress, PHIWA
  if constexpr (JAN) v4_sinput_jan<T, NANG, PP, 2, true, EXT>(tb, L, UFRIC, Z0M, RAORW, c[C_SIGN], coswdif, gx, gsp, xm0, xm1, wse, wslast, apl, rX, rY, sXY,
                                                                norma, sinwdif2, wp + 3 * NFRE, RNFAC);
  else if (norma) v4_sinput_n<T, NANG, PP, 2, true>(tb, L, wp + 3 * NFRE, UFRIC, Z0M, RAORW, RNFAC, c[C_SIGN], c[C_TEMP2], c[C_PTURB], c[C_AIRDPVISC], coswdif,
                                                   sinwdif2, gx, gsp, xm0, xm1, wse, wslast, apl, rX, rY, sXY);
  else v4_sinput<T, NANG, PP, 2, true>(tb, L, UFRIC, Z0M, RAORW, c[C_SIGN], c[C_TEMP2], c[C_PTURB], c[C_AIRDPVISC], sinwd, coswd, gx, gsp, xm0, xm1, wse,
                                       wslast, apl, rX, rY, sXY);
  femws_finish(wse, wslast, FMEANWS, EMW);
  MIJ = frcutindex4(FMEANWS, UFRIC);
  post_stress(MIJ, apl, true);
  WSYNC();
  } else {   // PART 2: what PART 1 handed over
    FMEANWS = c[C2_FMEANWS]; EMW = T(0);
    MIJ = (int)c[C_MIJ];
    MIJ = MIJ < 1 ? 1 : (MIJ > NFRE ? NFRE : MIJ);      // a table index: whatever the row holds, stay inside 1 .. NFRE
    SDS = c[C_SDS];
    if (tb.LWFLUX) {   // the XLLWS masks for FEMEANWS of the new spectrum: PART 1 wrote them to the point's XLLWS block
      const T* x0 = xllws + (size_t)ij * N + (size_t)(2 * j) * NFRE;
      for (int m = 0; m < NFRE; m++) {
        if (x0[m] != T(0)) xm0 |= (1ull << m);
        if (x0[NFRE + m] != T(0)) xm1 |= (1ull << m);
      }
    }
  }
  V4_PHASE_EXIT(205);
  // XLLWS(K,M) from the bit masks: rows K = 2j and 2j+1 of the point's block are contiguous
  auto store_xllws = [&]() {
    T* x0 = xllws + (size_t)ij * N + (size_t)(2 * j) * NFRE;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const unsigned long long xm = h ? xm1 : xm0;
#pragma unroll
      for (int m = 0; m < NFRE; m += VEC) {
        VT val;
#pragma unroll
        for (int i = 0; i < VEC; i++) val[i] = ((xm >> (m + i)) & 1ull) ? T(1) : T(0);
        *reinterpret_cast<VT*>(x0 + h * NFRE + m) = val;
      }
    }
  };
  if constexpr (PART == 1) {
    // ---- end of the first kernel: XLLWS, the scalars of the point for the second kernel and for k_implsch4_fin
    store_xllws();
    if (j == 0) {
      T* fr = fin + (size_t)ij * V4_NFIN;
      fr[FIN_AIRD] = AIRD; fr[FIN_UFRIC] = UFRIC; fr[FIN_Z0M] = Z0M; fr[FIN_MIJ] = c[C_MIJ];
      fr[FIN_XS] = c[C_XS]; fr[FIN_YS] = c[C_YS]; fr[FIN_F1DCOS3] = c[C_F1DCOS3]; fr[FIN_F1DCOS2] = c[C_F1DCOS2];
      fr[FIN_F1DSIN2] = c[C_F1DSIN2]; fr[FIN_F1D] = c[C_F1D]; fr[FIN_RNFAC] = c[C_RNFAC]; fr[FIN_PHIWA] = c[C_PHIWA];
      fr[FIN_SINWD] = sinwd; fr[FIN_COSWD] = coswd; fr[FIN_WSWAVE] = WSWAVE; fr[FIN_CICOVER] = CICOVER;
      fr[FIN_Z0B] = c[C_Z0B]; fr[FIN_CHRNCK] = c[C_CHRNCK];
      fr[FIN_EMEAN] = c[C_EMEAN]; fr[FIN_F1MEAN] = c[C_F1MEAN];
      fr[FIN_FMEAN] = FMEAN; fr[FIN_FMEANWS] = FMEANWS; fr[FIN_AKMEAN] = AKMEAN; fr[FIN_XKMEAN] = XKMEAN; fr[FIN_SDS] = SDS; fr[FIN_SC] = SCL;
      mij_out[ij] = MIJ;
    }
    return;
  }
  // ---- STRESSO of the second call (TAUW, TAUWDIR, PHIWA) and the scalar half of WNFLUXES: nothing below needs them, and on this
  //      kernel's waves they are a long dependent chain on PP lanes.  k_implsch4_fin runs them afterwards, one point per lane.
  V4_PHASE_EXIT(206);

  // ---- SDISSIP + SNONLIN + update sweep (implsch.F90:262-392), software-pipelined over the interaction frequencies MC = 1 .. MLSTHG:
  //   (1) LDS reads of the five rows interaction MC gathers from, of the row the dissipation is evaluated for (MC-4, one interaction
  //       ahead of its update) and of the factors of the row that is updated (MC-5): issued at the end of the previous interaction;
  //   (2) frequency-interpolated rows of the two quadruplet legs -> staging rows 0..3, rotated reads; saturation spectrum of row MC-4;
  //   (3) the DIA products of both mirror images -> staging rows 0..5, rotated reads;
  //   (4) increments into the register ring of eight rows; row MC-5 is complete: dissipation, limiter, new spectrum, fluxes.
  //   One LDS round trip between the stages; the directional maximum of the saturation spectrum (three dependent lane exchanges)
  //   rides along, one exchange per stage.  The coefficient record of the next interaction and the parked wind-input row come
  //   from global memory as vector loads one interaction (eight rows) ahead.
  V2<T> a_t = z2, a_x = z2;
  V2<T> a_ice = z2;      // RARE: integrand of the ice radiative stress
  {
    T ENHFR = m_max(T(0.75) * DEPTH * AKMEAN, T(0.5));
    ENHFR = T(1) + (T(5.5) / ENHFR) * (T(1) - T(0.833) * ENHFR) * m_exp(-T(1.25) * ENHFR);
    // ISNONLIN = 1 (snonlin.F90:138-150): ENH(MC) = MAX(MIN(ENH_MAX, TRANSF(k(MC), DEPTH)), ENH_MIN) per interaction frequency; lane j of
    // the point evaluates MC = q G + j + 1 (the wavenumber of the point's own table up to NFRE, the deep-water one beyond) and hands
    // it to the point's lanes with one lane exchange per interaction
    constexpr int NQE = ENHMC ? (V4_NFRE + 4 + G - 1) / G : 1;
    T rENH[NQE];
    int enh_lane0 = 0, enh_ext0 = 0;   // byte addresses of the point's lanes for pair 0 and for pair 16 (36 directions: the extras)
    if constexpr (ENHMC || RARE) {
      if constexpr (G == 18) { enh_lane0 = 4 * (16 * p); enh_ext0 = 4 * (48 + 2 * p - 16); }
      else { enh_lane0 = 4 * (p * G); enh_ext0 = enh_lane0; }
    }
    if constexpr (ENHMC) {
      // RARE: ISNONLIN = 2 -- the spectral widths of PEAK_ANG (peak_ang.F90:76-174) on the spectrum before any row is updated
      T XNU = T(0), SIG_TH = T(0);
      if constexpr (RARE) {
        if (tb.ISNONLIN == 2) {
          const T ZEPS = T(10) * (sizeof(T) == 4 ? T(1.1920928955078125e-07) : T(2.220446049250313e-16));   // 10 EPSILON
          const int NSHF = 1 + (int)(m_log(T(1.5)) / m_log(tb.FRATIO));
          const T* spt = sStg + SPOFF + p * NFRE;      // row totals over the directions of the point (the staging rows are idle here)
          for (int m = 0; m < NFRE; m++) {
            const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
            v4_row_total_to_lds<G, T>(f.x + f.y, L.rot, gsp, m);
          }
          WSYNC();
          V2<T> s01 = z2;
          T s2 = T(0);
#pragma unroll
          for (int q = 0; q < NS; q++) {
            const int m = mq(q, j);
            const T t2 = okq(q, j) ? spt[m] : T(0), fr = tb.FR[m], dfm = tb.DFIM[m];
            s01 = s01 + V2<T>{dfm, tb.DFIMFR[m]} * t2;
            s2 = s2 + (dfm * (fr * fr)) * t2;
          }
          s01 = v4_allsum<G, T>(s01, L.rot);
          s2 = v4_allsum1<G, T>(s2, L.rot);
          const T tl = spt[NFRE - 1];
          const T S0 = (ZEPS + s01.x) + (tb.WETAIL * frl * tb.DELTH) * tl;
          const T S1 = s01.y + (tb.WP1TAIL * tb.DELTH * (frl * frl)) * tl;
          const T S2 = s2 + (T(0.5) * tb.DELTH * (frl * frl * frl)) * tl;   // WP2TAIL = 0.5, yowfred.F90:54
          XNU = (S0 > ZEPS) ? m_sqrt(m_max(ZEPS, S2 * S0 / (S1 * S1) - T(1))) : ZEPS;
          // first maximum of F over M = 2 .. NFRE-1 in the reference's (M outer, K inner) order: the smallest M that attains it
          T vmax = T(0);
          int mm = 2;
          for (int m = 1; m < NFRE - 1; m++) {
            const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
            const T v = m_max(f.x, f.y);
            if (v > vmax) { vmax = v; mm = m + 1; }
          }
          const T gmax = v4_allmax1<G, T>(vmax, L.rot);
          const T cand = (gmax > T(0) && vmax == gmax) ? -T(mm) : -T(1.0e9);
          const T mneg = v4_allmax1<G, T>(cand, L.rot);
          const int MMAX = gmax > T(0) ? (int)(-mneg) : 2;      // per point
          const int MS = MMAX - NSHF > 1 ? MMAX - NSHF : 1, ME = MMAX + NSHF < NFRE ? MMAX + NSHF : NFRE;
          T SUM_S = T(0), SUM_C = ZEPS, SUM1 = ZEPS, SUM2 = T(0);
          const V2<T> thk = {tb.TH[2 * j], tb.TH[2 * j + 1]};
          for (int d = -NSHF; d <= NSHF; d++) {      // M = MMAX + d on every lane of a point, the rows outside MS .. ME contribute nothing
            const int M = MMAX + d;
            const bool in = (M >= MS && M <= ME);
            const int Mc = M < 1 ? 1 : (M > NFRE ? NFRE : M);
            V2<T> f = *reinterpret_cast<const V2<T>*>(tF + (Mc - 1) * RS);
            if (!in) f = z2;
            const V2<T> fs = f * L.sinth, fc = f * L.costh;
            const V2<T> a = v4_allsum<G, T>(V2<T>{fs.x + fs.y, fc.x + fc.y}, L.rot);
            SUM_S = SUM_S + a.x;
            SUM_C = SUM_C + a.y;
            const T THMEAN = m_atan2(SUM_S, SUM_C);
            const T dfim = tb.DFIM[Mc - 1];
            const V2<T> w = f * dfim;
            const V2<T> b = v4_allsum<G, T>(V2<T>{w.x + w.y, m_cos(thk.x - THMEAN) * w.x + m_cos(thk.y - THMEAN) * w.y}, L.rot);
            SUM1 = SUM1 + b.x;
            SUM2 = SUM2 + b.y;
          }
          SIG_TH = (SUM1 > ZEPS) ? m_sqrt(T(2) * (T(1) - SUM2 / SUM1)) : T(0);
          WSYNC();
        }
      }
#pragma unroll
      for (int q = 0; q < NQE; q++) {
        const int mc = q * G + j;   // MC - 1
        T XK;
        if (mc < NFRE) XK = L.fac4[mc * 4 + Q4_WAVNUM];
        else {
          T fr = T(1);
          for (int i = 0; i < mc + 1 - NFRE; i++) fr = fr * tb.FRATIO;
          const T w = tb.ZPIFR[NFRE - 1] * fr;
          XK = tb.GM1 * (w * w);
        }
        rENH[q] = m_max(m_min(T(10), transf_d(tb, XK, DEPTH)), T(0.1));
        if constexpr (RARE) {      // ISNONLIN decided at run time: 0 = the depth scaling from AKMEAN for every MC, 2 = TRANSF_SNL
          if (tb.ISNONLIN == 0) rENH[q] = ENHFR;
          if (tb.ISNONLIN == 2) rENH[q] = transf_snl_d(tb, XK, DEPTH, XNU, SIG_TH);
        }
      }
    }
    // a value the lanes of a point hold per frequency (lane j: rows q G + j) handed to all of them for row m: one lane exchange
    auto row_value = [&](const FLIV r, int m) -> T {
      if constexpr (!RARE) return T(0);
      else {
        const int q = m / G, jl = m - q * G;
        V4_CHK(m >= 0 && m < NFRE && q < NS);
        T v = r[0];
#pragma unroll
        for (int i = 1; i < NS; i++) v = (q == i) ? r[i] : v;
        const int base = (G == 18 && jl >= 16) ? enh_ext0 : enh_lane0;
        return v4_bp(base + 4 * jl, v);
      }
    };
    const bool ice2 = RARE && tb.LICERUN && tb.LCIWA2, wrs = RARE && tb.LWNEMOCOUWRS && tb.LCFLX;
    const bool ice13 = RARE && tb.LICERUN && (tb.LCIWA1 || tb.LCIWA3), ice3on = RARE && tb.LICERUN && tb.LCIWA3;
    const T EPSMIN1000 = tb.EPSMIN * T(1000);

    auto enh_of = [&](int MC) -> T {   // MC wave-uniform
      if constexpr (!ENHMC) return ENHFR;
      else {
        const int q = (MC - 1) / G, jl = (MC - 1) - q * G;
        V4_CHK(MC >= 1 && q < NQE);
        T v = rENH[0];
#pragma unroll
        for (int i = 1; i < NQE; i++) v = (q == i) ? rENH[i] : v;
        const int base = (G == 18 && jl >= 16) ? enh_ext0 : enh_lane0;
        return v4_bp(base + 4 * jl, v);
      }
    };
    const T DAL1 = tb.DAL1, DAL2 = tb.DAL2;
    const T CL11 = tb.DIAANG[0], ACL1 = tb.DIAANG[1], CL21 = tb.DIAANG[2], ACL2 = tb.DIAANG[3];
    const T CL11Q = tb.DIAANG[4], ACL1Q = tb.DIAANG[5], CL21Q = tb.DIAANG[6], ACL2Q = tb.DIAANG[7];
    constexpr bool RECPF_ON = (V4_RECPF != 0) && sizeof(T) == 4 && NANG <= 36 && !RARE;      // (see V4_RECPF)
    T wt[NH + 1];   // SATWEIGHTS depend on the tap only and are symmetric (checked by ecwam_hip_create): wave-uniform, taps -NH .. 0
#pragma unroll
    for (int t = 0; t <= NH; t++) wt[t] = v4_vreg<RECPF_ON>(tb.SATWEIGHTS[t][NANG / 2]);
    const T TMP03 = T(1) / (tb.SDSBR * tb.MICHE), SSDSC4 = tb.SSDSC4;
    const T c2 = tb.SSDSC2 * tb.SSDSC6, c2m1 = tb.SSDSC2 * (T(1) - tb.SSDSC6);
    const bool turb = !JAN && tb.SSDSC5 != T(0);
    const T FACTURB = turb ? (T(2) * tb.SSDSC5 / tb.G) * RAORW * UFRIC * UFRIC : T(0);
    const T DELT = (T)tb.IDELT, DELTM = T(1) / DELT, DELT5 = tb.XIMP * DELT;
    const bool shallow_brk = tb.LBIWBK && (DEPTH < T(50));
    const T USFM = UFRIC * m_max(FMEANWS, FMEAN);
    const T FSNL = (tb.LCFLX && tb.LWVFLX_SNL) ? T(1) : T(0);
    const T SDSL = shallow_brk ? SDS : T(0);
    const T BETA = (tb.LICERUN && tb.LCISCAL) ? T(1) - CICOVER : T(1);
    const int NRED = tb.NFRE_RED, MLST = tb.MLSTHG;
    T* st0 = sStg;            // staging rows: up / AD(kh=1), vp / DELAM(1), um / DELAP(1), vm / AD(2), row MC-4 / DELAM(2), DELAP(2)
    T* st1 = sStg + RS;
    T* st2 = sStg + 2 * RS;
    T* st3 = sStg + 3 * RS;
    T* st4 = sPl;
    T* st5 = sPl + PLN;
    V2<T> aS[8], aF[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { aS[i] = z2; aF[i] = z2; }
    // wind-input rows (parked in the XLLWS block by the second SINFLX call) come back through a ring of four prefetched rows:
    // slot jj & 3 holds row MC-5 while interaction MC is processed and is refilled with row MC-1 as soon as it is consumed
    V2<T> wiq[4];
#pragma unroll
    for (int i = 0; i < 4; i++) wiq[i] = *reinterpret_cast<const V2<T>*>(gx + (size_t)i * NANG);   // rows 0..3, first used by MC = 5..8
    // scalar clamps (a two-sided integer clamp of a uniform value is selected as v_med3_i32 and drags the address arithmetic into
    // the vector unit: signed maximum, then unsigned minimum)
    auto lo0 = [&](int r) { return r < 0 ? 0 : r; };
    auto hi35 = [&](int r) { return (int)((unsigned)r < (unsigned)(NFRE - 1) ? (unsigned)r : (unsigned)(NFRE - 1)); };
    // the un-updated tile rows MC-5 .. MC+2 (clamped to 0 .. NFRE-1: INLCOEF by formula, checked by ecwam_hip_create) stay in a register
    // ring, slot = row & 7: one LDS read per interaction (row MC+3, at the end of interaction MC into the slot of row MC-5) instead
    // of five.  Rows are updated in place behind the ring (row MC-5 at the end of interaction MC).
    V2<T> fR[8];
    {
      const V2<T> r0 = *reinterpret_cast<const V2<T>*>(tF);
      fR[4] = r0; fR[5] = r0; fR[6] = r0; fR[7] = r0; fR[0] = r0;
      fR[1] = *reinterpret_cast<const V2<T>*>(tF + RS);
      fR[2] = *reinterpret_cast<const V2<T>*>(tF + 2 * RS);
      fR[3] = *reinterpret_cast<const V2<T>*>(tF + 3 * RS);
    }
    // RHOWGDFTH(M) = RHOWG_DFIM(M) w(M), w = 1 below the cut-off MIJ, 1/2 at MIJ (1 if MIJ = NFRE), 0 above (frcutindex.F90:98-107):
    // w = clamp(MIJ + 1/2 - M, 0, 1); MIJ differs between the points of the wave
    const T MIJh = (T)MIJ + (MIJ == NFRE ? T(1) : T(0.5));
    // carried from one interaction to the next: saturation spectrum of row MC-4 with its directional maximum (last exchange in
    // flight) and that row's frequency; the record of the row the previous interaction completed (its update fills the wait for
    // the coefficient record and the first LDS round trip of this interaction)
    V2<T> bs_p = z2;
    T bm_p = T(0), e3_p = T(0), e4_p = T(0), sig_p = T(0);
    V2<T> u_f = z2, u_D = z2;
    T u_sbo = T(0), u_cinv = T(0), u_wn = T(0);
    // new spectrum, limiter and fluxes of row m (implsch.F90:300-392) from the finished ring slot and the parked wind input
    auto update_row = [&](int m, V2<T>& accS, V2<T>& accF, V2<T>& wslot, T cofr, T flmax, T rhowg, T sig) {
      const V2<T> f = u_f;
      V2<T> D = u_D;
      D = D - (sig * u_wn * FACTURB) * coswdif;   // (FACTURB = 0 without the turbulence term: exact, and cheaper than a uniform branch)
      const V2<T> fldw = D + wslot;
      V2<T> sl = fldw * f + accS;
      V2<T> fld = fldw + accF;
      V2<T> ss;
      {
        const V2<T> den = {m_max(T(1) - DELT5 * fld.x, T(1)), m_max(T(1) - DELT5 * fld.y, T(1))};
        ss = V2<T>{f_div(sl.x, den.x), f_div(sl.y, den.y)} * FSNL;   // FSNL = 1 / 0: LCFLX and LWVFLX_SNL
      }
      if constexpr (RARE) {
        if (tb.LCFLX && !tb.LWVFLX_SNL) ss = fldw * f;      // SL after SDISSIP, before SNONLIN, unmodulated (implsch.F90:280-288)
      }
      {
        const T sd = (m < NRED) ? SDSL : T(0);                        // SDIWBK where the point is shallow (SDSL = 0 elsewhere: exact)
        sl = sl - sd * f; fld = fld - sd;
      }
      sl = BETA * sl; fld = BETA * fld;                     // LCISCAL (implsch.F90:316-321); BETA = 1 without it: exact
      if constexpr (RARE) {
        if (ice2 || wrs) {
          V2<T> fli = z2;      // FLDICE of the last active SDICEn (each overwrites SLICE, sdice.F90:94-110)
          if (ice13) { const T v = row_value(rFLI, m); fli = V2<T>{v, v}; }
          if (ice2) {          // sdice2.F90:97-121: the attenuation grows with the wave height of the bin itself
            const T wn = u_wn, cgm = wp[NFRE + m];      // CGROUP(M) of the point: the factor table does not hold it
            const T dfm = tb.SINROW[m][1];
            const V2<T> EWH = {T(4) * m_sqrt(m_max(tb.EPSMIN, f.x * dfm)), T(4) * m_sqrt(m_max(tb.EPSMIN, f.y * dfm))};
            const V2<T> FL2 = -(((tb.CDICWA * (wn * wn)) * tb.ZALPFACB) * cgm) * EWH;
            sl = sl + CICOVER * (f * FL2); fld = fld + CICOVER * FL2;
            if (!ice3on) fli = FL2;
          }
          if (wrs) {           // SLICE = F FLDICE / GTEMP1 into the ice radiative stress integrand (wnfluxes.F90:178-196)
            const V2<T> num = f * fli;
            const V2<T> slice = {f_div(num.x, m_max(T(1) - DELT5 * fli.x, T(1))), f_div(num.y, m_max(T(1) - DELT5 * fli.y, T(1)))};
            a_ice = a_ice + (u_cinv * rhowg) * V2<T>{m_min(slice.x, -EPSMIN1000), m_min(slice.y, -EPSMIN1000)};
          }
        }
      }
      sl = sl + u_sbo * f; fld = fld + u_sbo;               // SBOTTOM (zero at M > NFRE_RED) + SDICE1 + SDICE3: the table slot
      const T lim = USFM * (cofr * DELT);
      V2<T> fn;
      {
        const T G0 = f_div_r(DELT * sl.x, m_max(T(1) - DELT5 * fld.x, T(1))), G1 = f_div_r(DELT * sl.y, m_max(T(1) - DELT5 * fld.y, T(1)));   // (refined: see f_div_r)
        fn.x = m_max(f.x + m_clamp_sym(G0, lim), FLM.x);
        fn.y = m_max(f.y + m_clamp_sym(G1, lim), FLM.y);
      }
      {
        // MIN(FLMAX - FL1, 0) is the capped value minus the uncapped one (the same subtraction where the cap bites, an exact zero elsewhere)
        const V2<T> fc = {m_min(fn.x, flmax), m_min(fn.y, flmax)};
        ss = ss + DELTM * (fc - fn);
        fn = fc;
      }
      *reinterpret_cast<V2<T>*>(tFw + m * RS) = fn;
      const T rh = rhowg * m_min(m_max(MIJh - (T)(m + 1), T(0)), T(1));
      a_t = a_t + rh * ss;
      a_x = a_x + (u_cinv * rh) * ss;
      // the slot of the wind-input ring is free: row m+4 (beyond the last row the last one again, never used: no branch)
      V4_CHK(m >= 0 && m < NFRE && ij >= kijs && ij < kijl);
      wslot = *reinterpret_cast<const V2<T>*>(gx + (size_t)hi35(m + 4) * NANG);
    };
    if constexpr (!JAN) *reinterpret_cast<V2<T>*>(st4 + L.own) = fR[5];   // row MC-4 of the first interaction
    // WIN: the frequency-interpolated rows W+ / W- of an interaction are staged at the END of the interaction before it (their write -> read
    // round trip then hides behind the next interaction's coefficient loads and row update instead of standing in its critical path);
    // here those of the first interaction
    V2<T> wp_c = z2, wm_c = z2;
    {
      const T* cw0 = tb.DIAW[0];
      wp_c = cw0[0] * fR[2] + cw0[1] * fR[3];
      wm_c = cw0[2] * fR[4] + cw0[3] * fR[5];
      *reinterpret_cast<V2<T>*>(st0 + L.own) = wp_c; *reinterpret_cast<V2<T>*>(st2 + L.own) = wm_c;
    }
    V4SYNC();
    // Rows above the cut-off MIJ need no source terms: IMPHFTAIL replaces them by the tail of row MIJ (imphftail.F90:77-88), their flux
    // weights RHOWGDFTH are zero (frcutindex.F90:100-107), and the only other reader of the updated rows before the tail goes in --
    // FEMEANWS of implsch.F90:427 -- feeds nothing but the LWFLUX block (:435-445).  Without LWFLUX (and LWNEMOCOUWRS) the sweep therefore ends with the
    // highest cut-off of the wave's points: interaction MC adds to the rows MC-4 .. MC+3, so the last one needed is MIJ + 4, and the last
    // row updated is row MIJ.  (Exact: the same bits in every output.  On sea states whose cut-off lies in the middle of the frequency
    // range that is a third of the interactions; under sea ice and in light winds MIJ = NFRE and nothing is skipped.)
    int mijmax = 0;
#pragma unroll
    for (int q = 0; q < PP; q++) {
      const int mq = __builtin_amdgcn_readlane(MIJ, G == 18 ? 16 * q : q * G);
      mijmax = mq > mijmax ? mq : mijmax;
    }
    const bool whole = tb.LWFLUX != 0 || wrs;      // (the ice radiative stress integrates SLICE over every row, wnfluxes.F90:178-196)
    const int UPD_LIM = whole ? NFRE : mijmax;                                      // rows m < UPD_LIM (0-based) are updated
    const int DIA_LIM = whole ? MLST : (mijmax + 4 < MLST ? mijmax + 4 : MLST);     // interactions MC <= DIA_LIM contribute to them
    const int MC_END = whole ? MLST : (mijmax + 5 < MLST ? mijmax + 5 : MLST);      // row MIJ is updated at the top of interaction MIJ + 5
    // V4_RECV = 2 (double precision): the record of THIS interaction by scalar loads issued together at its top and held there (one
    // register set: two are 76 scalar registers) -- one wait per interaction instead of six
    constexpr bool RECS = !RARE && (V4_RECV == 2) && sizeof(T) == 8;      // (sp at 48 directions: its spilled scalar registers push the vector registers past 256)
    constexpr bool RECPF = (RECPF_ON || RECS);
    T ra[20], rb[20];      // RECPF: the records of two consecutive interactions, roles alternating (the loop is unrolled by eight: static)
    if constexpr (RECPF) {
      const T* rp0 = tb.DIAREC[0];
      if constexpr (!RECS) {
#pragma unroll
        for (int i = 0; i < 20; i++) ra[i] = rp0[i];
      }
      if constexpr (!RECS) { v4_pin10(ra, 0); v4_pin10(ra, 10); }
    }
    int MCb = 0;
    for (; MCb < MC_END; MCb += 8) {
#pragma unroll
      for (int jj = 0; jj < 8; jj++) {
        const int MC = MCb + 1 + jj;
        T (&rc)[20] = (RECS || !(jj & 1)) ? ra : rb;      // this interaction's record
        T (&rn)[20] = (jj & 1) ? ra : rb;      // the next one's
        if constexpr (RECS) {
          const T* rp = tb.DIAREC[MC - 1];
#pragma unroll
          for (int i = 0; i < 20; i++) ra[i] = rp[i];
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (RECPF && !RECS) {      // the record of interaction MC + 1 (row MLSTHG of the table repeats the last one)
          const T* rp = tb.DIAREC[MC < MLST ? MC : MLST];
#pragma unroll
          for (int i = 0; i < 20; i++) rn[i] = rp[i];
        }
        const int c0 = (1 + jj) & 7, cm = (1 + jj + 4) & 7, cm1 = (1 + jj + 5) & 7, cp = (1 + jj + 2) & 7, cp1 = (1 + jj + 3) & 7;
        const int m = MC - 5;   // row that is complete after this interaction (updated at the top of the next one)
        // ---- stage 1: row MC-4 back from its staging row (rotated reads at fixed addresses), the factors of rows MC-5 and MC-4
        const V2<T> fIC = fR[jj & 7], fIP = fR[(jj + 2) & 7], fIM = fR[(jj + 4) & 7], fIM1 = fR[(jj + 5) & 7];
        T el[4 * NSH + 2];                // row MC-4: element e = F(2j - 2 NSH + e)
        V2<T> elp[2 * NSH + 1];           // the same as aligned pairs
        const int IM = lo0(MC - 5), IM1 = hi35(lo0(MC - 4));
        V4_CHK(MC >= 1 && MC <= 40 && IM >= 0 && IM < NFRE && IM1 >= 0 && IM1 < NFRE);
        T bscn = T(0);
        if constexpr (!JAN) {
#pragma unroll
          for (int i = 0; i <= 2 * NSH; i++) {
            const V2<T> v = (i == NSH) ? fIM1 : *reinterpret_cast<const V2<T>*>(st4 + sh[i]);
            el[2 * i] = v.x; el[2 * i + 1] = v.y; elp[i] = v;
          }
          bscn = L.fac4[IM1 * 4 + Q4_BSC];
        }
        const V2<T> qf0 = *reinterpret_cast<const V2<T>*>(L.fac4 + IM * 4), qf1 = *reinterpret_cast<const V2<T>*>(L.fac4 + IM * 4 + 2);
        // WIN: row MC+3 enters the ring now, in the slot of row MC-4 (fIM above holds it: the rows of this interaction were staged by the
        // previous one); it is IP1 of the next interaction, whose rows are staged at the end of this one
        fR[(jj + 4) & 7] = *reinterpret_cast<const V2<T>*>(tF + hi35(MC + 3) * RS);
        // ---- coefficient record of the interaction (wave-uniform)
        const T* cg = tb.DIACF[MC - 1];
        const T* cs = cg + 12;
        const T GW1 = cg[1], GW2 = cg[2], GW3 = cg[3], GW4 = cg[4], GW5 = cg[5], GW6 = cg[6], GW7 = cg[7], GW8 = cg[8];
        const T* cw = RECPF ? rc + 4 : tb.DIAW[MC - 1];      // the separable form (WIN): frequency factors of the gathers and of the scatter (words 4 .. 15)
        const T cg9 = RECPF ? rc[0] : cg[9], cg10 = RECPF ? rc[1] : cg[10], cg11 = RECPF ? rc[2] : cg[11];
        const T cg28 = RECPF ? rc[3] : cg[28], cg29 = RECPF ? rc[4] : cg[29], cg30 = RECPF ? rc[5] : cg[30], cg31 = RECPF ? rc[6] : cg[31];
        const T FTEMP = cg9 * enh_of(MC <= MLST ? MC : MLST);
        const T FKLAMPA = cs[0], FKLAMPB = cs[1], FKLAMP2 = cs[2], FKLAMP1 = cs[3];
        const T FKLAPA2 = cs[4], FKLAPB2 = cs[5], FKLAP12 = cs[6], FKLAP22 = cs[7];
        const T FKLAMMA = cs[8], FKLAMMB = cs[9], FKLAMM2 = cs[10], FKLAMM1 = cs[11];
        const T FKLAMA2 = cs[12], FKLAMB2 = cs[13], FKLAM12 = cs[14], FKLAM22 = cs[15];
        const T FTAIL = cg31;   // the tail factor RNLCOEF(1), or 1 between MFR1STFR and MFRLSTFR where the reference skips it (x 1 is exact)
        // ---- meanwhile: the row the previous interaction completed
        if (m - 1 >= 0 && m - 1 < UPD_LIM) update_row(m - 1, aS[(jj + 4) & 7], aF[(jj + 4) & 7], wiq[(jj + 3) & 3], cg28, cg29, cg30, cg11);
        // the dissipation coefficient of row m (sdissip_ard.F90:117-314) from the saturation spectrum and the maximum the previous
        // interaction left in flight
        if constexpr (JAN) {   // SDISSIP_JAN: the rate of row m sits in the saturation slot of the factor table
          u_D = V2<T>{qf0.x, qf0.x};
          u_f = fIM; u_sbo = qf0.y; u_cinv = qf1.x; u_wn = qf1.y;
        } else {
          const T bm = (G == 18) ? e3_p : m_max(bm_p, m_max(e3_p, e4_p));
          const T d0 = m_max(T(0), bm * TMP03 - SSDSC4);
          const V2<T> t1 = bs_p * TMP03 - SSDSC4;
          const V2<T> d1 = {m_max(T(0), t1.x), m_max(T(0), t1.y)};
          u_D = (c2 * sig_p) * (d0 * d0) + (c2m1 * sig_p) * (d1 * d1);
          u_f = fIM; u_sbo = qf0.y; u_cinv = qf1.x; u_wn = qf1.y;
        }
        // ---- stage 2: frequency-interpolated rows of the + and - legs (snonlin.F90:236-262) -> staging rows, rotated reads; meanwhile
        //      the saturation spectrum of row MC-4 (SATWEIGHTS symmetric about the centre tap) and the first exchange of its maximum
        const V2<T> FIJ = fIC * FTAIL;
        const V2<T> fIP1 = fR[(jj + 3) & 7];
        V2<T> SAPk[2], SAMk[2];
        {
          // W+ = GP F(:,IP) + GP1 F(:,IP1), W- = GM F(:,IM) + GM1 F(:,IM1): two staged rows; SAP = CL11 W+(K1) + ACL1 W+(K11), SAM likewise
          // (staged by the previous interaction; wp_c / wm_c are the lane's own pairs of the two rows)
          const V2<T> wp = wp_c, wm = wm_c;
          // kh = 1: K1 = K - R1, K11 = K - R1 - 1, K2 = K + R2, K21 = K + R2 + 1; kh = 2 mirrored
          SAPk[0] = v4_win<T, NSH, -R1, -(R1 + 1)>(st0, sh, wp, CL11, ACL1);
          SAMk[0] = v4_win<T, NSH, R2, R2 + 1>(st2, sh, wm, CL21, ACL2);
          SAPk[1] = v4_win<T, NSH, R1, R1 + 1>(st0, sh, wp, CL11, ACL1);
          SAMk[1] = v4_win<T, NSH, -R2, -(R2 + 1)>(st2, sh, wm, CL21, ACL2);
        }
        V2<T> bsat = z2;
        T bm1 = T(0), e0 = T(0);
        if constexpr (!JAN) {
          bsat = V2<T>{wt[NH] * el[2 * NSH], wt[NH] * el[2 * NSH + 1]};
#pragma unroll
          for (int d = 1; d <= NH; d++) {
            bsat.x += wt[NH - d] * (el[2 * NSH - d] + el[2 * NSH + d]);
            bsat.y += wt[NH - d] * (el[2 * NSH + 1 - d] + el[2 * NSH + 1 + d]);
          }
          bsat = bsat * bscn;
          bm1 = m_max(bsat.x, bsat.y);
          e0 = v4_bp(L.rot.a0, bm1);
        }
        V4SYNC();
        // ---- stage 3: the DIA products of the two mirror images (snonlin.F90:264-306)
        const V2<T> FCEN = FTEMP * FIJ;
        const V2<T> FCD1 = DAL1 * FCEN, FCD2 = DAL2 * FCEN;   // shared by the two mirror images
        T e1 = T(0), e2 = T(0), e3 = T(0), e4 = T(0);
        // (the test -- true for every interaction when nothing is cut off -- also splits the basic block: scheduled as one block, the eight
        // unrolled interactions need 340 VGPRs)
        // The two mirror images go through separate staging rows (0..2 and 3..5) so that the reads of the first are in flight while
        // the products of the second are formed, and the reads of the second while the increments of the first are added.
        if (MC <= DIA_LIM) {
          V2<T> ADk[2], DELADk[2];
          V2<T> A2[2], A2s[2], A1[2], A1s[2], D2[2], D2s[2], P1[2], P1s[2];
#pragma unroll
          for (int kh = 0; kh < 2; kh++) {
            const V2<T> SAP = SAPk[kh], SAM = SAMk[kh];
            V2<T> FAD1 = FIJ * (SAP + SAM);
            const V2<T> FAD2 = FAD1 - T(2) * SAP * SAM;
            FAD1 = FAD1 + FAD2;
            const V2<T> AD = FAD2 * FCEN;
            ADk[kh] = AD;
            DELADk[kh] = FAD1;      // (times FTEMP below, on the sum of the two mirror images)
            const V2<T> DELAP = (FIJ - T(2) * SAM) * FCD1;
            const V2<T> DELAM = (FIJ - T(2) * SAP) * FCD2;
            T* sa = kh == 0 ? st0 : st3;
            T* sm = kh == 0 ? st1 : st4;
            T* sp = kh == 0 ? st2 : st5;
            *reinterpret_cast<V2<T>*>(sa + L.own) = AD; *reinterpret_cast<V2<T>*>(sm + L.own) = DELAM;
            *reinterpret_cast<V2<T>*>(sp + L.own) = DELAP;
            V4SYNC();
            if (kh == 0) {
              {      // the angular interpolation of the scatter: one window per (quantity, leg)
                A2[0] = v4_win<T, NSH, -R2, -(R2 + 1)>(sa, sh, AD, CL21, ACL2); A1[0] = v4_win<T, NSH, R1, R1 + 1>(sa, sh, AD, CL11, ACL1);
                D2[0] = v4_win<T, NSH, -R2, -(R2 + 1)>(sm, sh, DELAM, CL21Q, ACL2Q); P1[0] = v4_win<T, NSH, R1, R1 + 1>(sp, sh, DELAP, CL11Q, ACL1Q);
              }
              if constexpr (G != 18 && !JAN) {
                bm1 = m_max(bm1, e0);
                if (G >= 12) e1 = v4_bp(L.rot.a1, bm1);
              }
            } else {
              {
                A2[1] = v4_win<T, NSH, R2, R2 + 1>(sa, sh, AD, CL21, ACL2); A1[1] = v4_win<T, NSH, -R1, -(R1 + 1)>(sa, sh, AD, CL11, ACL1);
                D2[1] = v4_win<T, NSH, R2, R2 + 1>(sm, sh, DELAM, CL21Q, ACL2Q); P1[1] = v4_win<T, NSH, -R1, -(R1 + 1)>(sp, sh, DELAP, CL11Q, ACL1Q);
              }
              if constexpr (JAN) {
              } else if constexpr (G == 18) {   // extras folded in, row maximum (no LDS), back to the extras: read by the next interaction
                bm1 = v4_rowmax<T>(m_max(bm1, e0));
                e3 = v4_bp(L.rot.a1, bm1);
              } else {
                if (G >= 12) bm1 = m_max(bm1, e1);
                if (G == 24) bm1 = m_max(bm1, v4_bp(L.rot.a2, bm1));      // (48 directions: one more exchange, waited for here)
                e3 = v4_bp(L.rot.a3, bm1); e4 = v4_bp(L.rot.a4, bm1);
              }
            }
          }
          V4SYNC();
          {
            // the two mirror images carry the same coefficients: their products are added first, then one fused multiply-add per
            // coefficient (the reference adds the terms of KH = 1 and KH = 2 one after the other: the same sum in another order)
            const V2<T> ADt = ADk[0] + ADk[1], DELADt = DELADk[0] + DELADk[1];
            aS[c0] -= T(2) * ADt;
            aF[c0] -= (T(2) * FTEMP) * DELADt;
            {
              // the interpolated increments of the two mirror images, one fused multiply-add per target row with the frequency factor
              const V2<T> A2t = A2[0] + A2[1], A1t = A1[0] + A1[1], D2t = D2[0] + D2[1], P1t = P1[0] + P1[1];
              aS[cm] += A2t * cw[9]; aF[cm] += D2t * cw[11];        // FKLAMM1, its square
              aS[cm1] += A2t * cw[8]; aF[cm1] += D2t * cw[10];      // FKLAMM, its square
              aS[cp] += A1t * cw[5]; aF[cp] += P1t * cw[7];         // FKLAMP1, its square
              aS[cp1] = A1t * cw[4]; aF[cp1] = P1t * cw[6];         // FKLAMP, its square: the first contribution to the row that entered the ring
            }
          }
        }
        // ---- row MC-3 (the saturation row of the next interaction) -> its staging row; row MC+3 enters the ring in the slot of row MC-5
        if constexpr (!JAN) *reinterpret_cast<V2<T>*>(st4 + L.own) = fR[(jj + 6) & 7];
        {      // W+ / W- of interaction MC + 1: IP = row MC+2, IP1 = MC+3 (entered the ring above), IM = MC-3, IM1 = MC-2
          wp_c = cw[12] * fR[(jj + 3) & 7] + cw[13] * fR[(jj + 4) & 7];
          wm_c = cw[14] * fR[(jj + 5) & 7] + cw[15] * fR[(jj + 6) & 7];
          *reinterpret_cast<V2<T>*>(st0 + L.own) = wp_c; *reinterpret_cast<V2<T>*>(st2 + L.own) = wm_c;
        }
        bs_p = bsat; bm_p = bm1; e3_p = e3; e4_p = e4; sig_p = cg10;
        if constexpr (RECPF && !RECS) {      // the next record is complete by now: pinned (not re-loaded piecemeal), it becomes the current one
          v4_pin10(rn, 0); v4_pin10(rn, 10);
        }
        V4SYNC();
      }
    }
    // the row the last interaction completed (MCb is a multiple of 8 here: static ring slots)
    if (MCb - 5 >= 0 && MCb - 5 < UPD_LIM)
      update_row(MCb - 5, aS[4], aF[4], wiq[3], lane_get(L.rCOFRM4, MCb - 5), lane_get(L.rFLMAX, MCb - 5), lane_get(L.rRHOWG, MCb - 5), lane_get(L.rZPIFR, MCb - 5));
  }
  V4SYNC();
  V4_PHASE_EXIT(207);
  const T Z0B = c[C_Z0B], CHRNCK = c[C_CHRNCK];
  // the Stokes-drift weights STOKFAC(M) DFIM_SIM(M) of the lane's frequencies: loaded here, used behind the flux sums (their
  // round trip to memory hides behind the three all-reduces)
  T stkw[NS], stkd[NS];
  T xk2r[JAN ? NS : 1];
#pragma unroll
  for (int q = 0; q < NS; q++) {   // unconditional loads, pinned here (a conditional one is sunk into a branch behind the sums: load, wait, twice)
    const int m = mq(q, j);
    stkw[q] = wp[4 * NFRE + m];
    stkd[q] = tb.DFIM_SIM[m];
    if constexpr (JAN) xk2r[q] = wp[3 * NFRE + m];   // XK2CG again: the saturation scale WAVNUM XK2CG / 2 pi back into its slot for IMPHFTAIL
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- WNFLUXES (wnfluxes.F90:147-190): the directional sums of the flux accumulators; the rest of it in k_implsch4_fin
  T PHILF = T(0), XSTRESS = T(0), YSTRESS = T(0);
  if (tb.LCFLX) {
    const V2<T> sx = a_x * L.sinth, sy = a_x * L.costh;
    const V2<T> r0 = v4_allsum<G, T>(V2<T>{a_t.x + a_t.y, sx.x + sx.y}, L.rot);
    YSTRESS = v4_allsum1<G, T>(sy.x + sy.y, L.rot);
    PHILF = r0.x; XSTRESS = r0.y;
  }
  T TAUICX = T(0), TAUICY = T(0);
  if constexpr (RARE) {
    if (tb.LCFLX && tb.LWNEMOCOUWRS) {   // wnfluxes.F90:178-196, 267-271: the stress on the ice, sign flipped
      const V2<T> ix = a_ice * L.sinth, iy = a_ice * L.costh;
      const V2<T> r1 = v4_allsum<G, T>(V2<T>{ix.x + ix.y, iy.x + iy.y}, L.rot);
      TAUICX = -(tb.ZALPWRS * r1.x); TAUICY = -(tb.ZALPWRS * r1.y);
    }
  }

  // ---- second FEMEANWS, IMPHFTAIL, SETICE, STOKESDRIFT (implsch.F90:422-462).  The second FKMEAN (implsch.F90:422) has no reader: its
  //      five means are locals of IMPLSCH that nothing uses after it (WNFLUXES ran before it on the means of the old spectrum, MIJ is
  //      not recomputed); FEMEANWS of the new spectrum is read by the LWFLUX block only (implsch.F90:435-445).
  //      The Stokes-drift weights STOKFAC(M) DFIM_SIM(M) of the point go to the LOG plane now: read inside the row loop below
  //      they were one exposed global-memory round trip per frequency.
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const int m = mq(q, j);
    L.zcn[m] = (m < tb.NFRE_ODD) ? stkw[q] * stkd[q] : T(0);
    if constexpr (JAN) L.fac4[m * 4 + Q4_BSC] = L.fac4[m * 4 + Q4_WAVNUM] * (T(1) / tb.ZPI) * xk2r[q];
  }
  WSYNC();
  T EMEANWS = T(0);
  if (tb.LWFLUX) {  // femeanws.F90:84-123 on the new spectrum, before the tail is replaced
    V2<T> we = z2, wl = z2;   // windsea (EM, FM), windsea part of the last row
    for (int m = 0; m < NFRE; m++) {
      const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
      const T* row = tb.SINROW[m];
      const V2<T> x = {((xm0 >> m) & 1ull) ? f.x : T(0), ((xm1 >> m) & 1ull) ? f.y : T(0)};
      we = we + V2<T>{RLANE ? lane_get(L.rDFIM, m) : row[1], RLANE ? lane_get(L.rDFIMOFR, m) : row[5]} * (x.x + x.y);
      wl = x;
    }
    femws_finish(we, wl, FMEANWS, EMEANWS);
  }
  V4_PHASE_EXIT(208);
  {  // imphftail.F90: TEMP2(M) / TEMP1 = (XK2CG WAVNUM)(MIJ) / (XK2CG WAVNUM)(M)
    V4_CHK(MIJ >= 1 && MIJ <= NFRE);
    const T B1 = L.fac4[(MIJ - 1) * 4 + Q4_BSC];
    const V2<T> tf = *reinterpret_cast<const V2<T>*>(tF + (MIJ - 1) * RS);
    for (int m = MIJ; m < NFRE; m++) {
      const T tm = B1 / L.fac4[m * 4 + Q4_BSC];
      *reinterpret_cast<V2<T>*>(tFw + m * RS) = V2<T>{m_max(tm * tf.x, FLM.x), m_max(tm * tf.y, FLM.y)};
    }
  }
  if (tb.LICERUN && tb.LMASKICE) {  // setice.F90:67-86
    T CIREDUC, ICEFREE;
    if (CICOVER > tb.CITHRSH) { CIREDUC = m_max(tb.EPSMIN, T(1) - CICOVER); ICEFREE = T(0); }
    else { CIREDUC = T(0); ICEFREE = T(1); }
    const V2<T> add = (CIREDUC * tb.FLMIN) * (cpos * cpos);
    for (int m = 0; m < NFRE; m++) *reinterpret_cast<V2<T>*>(tFw + m * RS) = *reinterpret_cast<const V2<T>*>(tF + m * RS) * ICEFREE + add;
  }
  T USTOKES, VSTOKES;
  {  // stokesdrift.F90:89-142
    const int MO = tb.NFRE_ODD;
    const T fo = tb.FR[MO - 1];
    const T CONST = T(2) * tb.DELTH * (tb.ZPI * tb.ZPI * tb.ZPI) / tb.G * m_pow4(fo);
    V2<T> a = z2;
    for (int m = 0; m < MO; m++) a = a + L.zcn[m] * *reinterpret_cast<const V2<T>*>(tF + m * RS);
    a = a + CONST * *reinterpret_cast<const V2<T>*>(tF + (MO - 1) * RS);
    const V2<T> ax = a * L.sinth, ay = a * L.costh;
    const V2<T> s = v4_allsum<G, T>(V2<T>{ax.x + ax.y, ay.x + ay.y}, L.rot);
    USTOKES = s.x; VSTOKES = s.y;
    if (tb.LICERUN && tb.LWAMRSETCI && CICOVER > tb.CITHRSH) {
      USTOKES = T(0.016) * WSWAVE * sinwd * (T(1) - CICOVER);
      VSTOKES = T(0.016) * WSWAVE * coswd * (T(1) - CICOVER);
    }
    USTOKES = m_min(m_max(USTOKES, T(-1.5)), T(1.5));
    VSTOKES = m_min(m_max(VSTOKES, T(-1.5)), T(1.5));
  }
  T STRNMS = T(0);
  if constexpr (RARE) {
    if (tb.LWNEMOCOUSTRN) {   // cimsstrn.F90:86-118 with aki_ice.F90:60-112 on the new spectrum: lane j takes the frequencies q G + j + 1
      WSYNC();
      const T* spt = sStg + SPOFF + p * NFRE;      // row totals over the directions (the staging rows are idle again)
      for (int m = 0; m < NFRE; m++) {
        const V2<T> f = *reinterpret_cast<const V2<T>*>(tF + m * RS);
        v4_row_total_to_lds<G, T>(f.x + f.y, L.rot, gsp, m);
      }
      WSYNC();
      T term = T(0);
#pragma unroll
      for (int q = 0; q < NS; q++) {
        const int m = mq(q, j);
        const T wn = L.fac4[m * 4 + Q4_WAVNUM];
        const T XKI = aki_ice_d(tb.G, wn, DEPTH, tb.ROWATER, CITHICKi);
        const T E = T(0.5) * CITHICKi * (XKI * XKI * XKI) / wn;
        const T sume = spt[m];
        if (okq(q, j) && sume > tb.FLMIN / tb.DELTH) term += (E * E) * sume * tb.DFIM[m];
      }
      STRNMS = v4_allsum1<G, T>(term, L.rot);
    }
  }
  WSYNC();
  V4_PHASE_EXIT(209);
  // ---- store FL1 (16-byte chunks gathered from VEC rows of the tile), XLLWS(K,M) from the bit masks, per-point scalars
  {
    constexpr int NV = N / VEC, NIT = (NV + 63) / 64;
    int k = lane / NC, r = lane - k * NC;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int w = lane + 64 * it;
      if (w < NV) {
        const T* d = sT + (r * VEC) * RS + rowpos(0, k);
        const int qo = NANG;
#pragma unroll
        for (int q = 0; q < PP; q++) {
          if (q < n) {
            VT val;
#pragma unroll
            for (int i = 0; i < VEC; i++) val[i] = d[i * RS + q * qo];
            *reinterpret_cast<VT*>(fl1 + (size_t)(ij0 + q) * N + (size_t)w * VEC) = val;
            if (gfast && r * VEC < gk) *reinterpret_cast<VT*>(gfast + ((size_t)(ij0 + q) * NANG + k) * gk + r * VEC) = val;
          }
        }
      }
      k += 64 / NC; r += 64 % NC;
      if (r >= NC) { r -= NC; k += 1; }
    }
  }
  V4_PHASE_EXIT(210);
  if constexpr (PART == 0) store_xllws();   // every wind-input row parked in this block has been read by now
  V4_PHASE_EXIT(211);
  if (j == 0) {
    T* fo = ffa + (size_t)ij * ECWAM_HIP_NFF;
    fo[7] = UFRIC; fo[10] = Z0M; fo[11] = Z0B; fo[12] = CHRNCK;   // TAUW, TAUWDIR: k_implsch4_fin
    if (RARE && tb.ICODE != 3) fo[3] = WSWAVE;                    // friction-velocity forcing: the 10 m wind is an output (airsea.F90:107-115)
    T* io = intfa + (size_t)ij * ECWAM_HIP_NINTF;
    io[2] = USTOKES; io[3] = VSTOKES;
    T* fr = fin + (size_t)ij * V4_NFIN;
    if constexpr (PART == 0) {   // (PART 2: the first kernel wrote these)
      fr[FIN_AIRD] = AIRD; fr[FIN_UFRIC] = UFRIC; fr[FIN_Z0M] = Z0M; fr[FIN_MIJ] = c[C_MIJ];
      fr[FIN_XS] = c[C_XS]; fr[FIN_YS] = c[C_YS]; fr[FIN_F1DCOS3] = c[C_F1DCOS3]; fr[FIN_F1DCOS2] = c[C_F1DCOS2];
      fr[FIN_F1DSIN2] = c[C_F1DSIN2]; fr[FIN_F1D] = c[C_F1D]; fr[FIN_RNFAC] = c[C_RNFAC]; fr[FIN_PHIWA] = c[C_PHIWA];
      fr[FIN_SINWD] = sinwd; fr[FIN_COSWD] = coswd; fr[FIN_WSWAVE] = WSWAVE; fr[FIN_CICOVER] = CICOVER;
      fr[FIN_EMEAN] = c[C_EMEAN]; fr[FIN_F1MEAN] = c[C_F1MEAN];   // of the spectrum before the update (implsch.F90:396-414)
    }
    fr[FIN_PHILF] = PHILF; fr[FIN_XSTRESS] = XSTRESS; fr[FIN_YSTRESS] = YSTRESS;
    fr[FIN_TAUICX] = TAUICX; fr[FIN_TAUICY] = TAUICY; fr[FIN_STRNMS] = STRNMS;
    if (RARE && tb.LWNEMOCOUSTRN) io[4] = STRNMS;
    if (tb.LWFLUX) {
      io[0] = (EMEANWS < tb.WSEMEAN_MIN) ? tb.WSEMEAN_MIN : EMEANWS;
      io[1] = (EMEANWS < tb.WSEMEAN_MIN) ? T(2) * tb.FR[NFRE - 1] : FMEANWS;
    }
    mij_out[ij] = MIJ;
  }
}

// The scalar start of the time step, one sea point per lane (sinflx.F90:105-122): direction of the wind, RNFAC, the first TAUT_Z0
// (taut_z0.F90:288-340; with LLGCBZ0 only its COSDIFF -- the gravity-capillary iteration needs the spectrum and stays in k_implsch4).
template <typename T, bool EXT, bool RARE = false>
__global__ void __launch_bounds__(64) k_implsch4_pre(const DevTab<T>* __restrict__ tp, int kijs, int kijl, const T* __restrict__ ffa,
                                                     T* __restrict__ fin) {
  const DevTab<T>& tb = *tp;
  const int ij = kijs + blockIdx.x * 64 + threadIdx.x;
  if (ij >= kijl) return;
  const T* ff = ffa + (size_t)ij * ECWAM_HIP_NFF;
  T* fr = fin + (size_t)ij * V4_NFIN;
  const T WDWAVE = ff[1], WSWAVE = ff[3];
  fr[FIN_SINWD] = m_sin(WDWAVE); fr[FIN_COSWD] = m_cos(WDWAVE);
  T RNFAC = T(1);   // sinflx.F90:116-120
  if (EXT && tb.LLNORMAGAM && tb.LLCAPCHNK) RNFAC = T(1) + tb.DTHRN_A * (T(1) + m_tanh(WSWAVE - tb.DTHRN_U));
  fr[FIN_RNFAC] = RNFAC;
  T UFRIC = ff[7], Z0M = ff[10], Z0B = ff[11], CHRNCK = ff[12];
  if (EXT && tb.LLGCBZ0) fr[FIN_COSDIFF] = m_cos(WDWAVE - ff[9]);
  if (RARE && tb.ICODE != 3) {
    // friction-velocity forcing (airsea.F90:100-117): Z0WAVE (z0wave.F90:73-92), then the 10 m wind from the log profile
    const T ALPHAOG = (tb.LLCAPCHNK ? chnkmin(tb, WSWAVE) : tb.ALPHA) * tb.GM1;
    const T UST2 = UFRIC * UFRIC, UST3 = UST2 * UFRIC;
    const T ARG = m_max(UST2 - ff[8], tb.EPS1);
    Z0M = ALPHAOG * UST3 / m_sqrt(ARG);
    Z0B = ALPHAOG * UST2;
    CHRNCK = tb.G * Z0M / UST2;
    fr[FIN_WSWAVE] = m_max((T(1) / tb.XKAPPA) * UFRIC * (m_log(tb.XNLEV) - m_log(Z0M)), tb.WSPMIN);
  } else if (!(EXT && tb.LLGCBZ0)) taut_z0_a(tb, 0, WSWAVE, WDWAVE, ff[8], ff[9], UFRIC, Z0M, Z0B, CHRNCK);
  fr[FIN_UFRIC] = UFRIC; fr[FIN_Z0M] = Z0M; fr[FIN_Z0B] = Z0B; fr[FIN_CHRNCK] = CHRNCK;
}

// The scalar end of the time step, one sea point per lane: STRESSO's second call (stresso.F90:180-229 with tau_phi_hf.F90:125-301:
// TAUW, TAUWDIR, PHIWA) and WNFLUXES' point-wise part (wnfluxes.F90:190-330, LWNEMOCOU = F) from the row of scalars k_implsch4 left
// in fin(:, IJ).  On k_implsch4's waves these two dependent chains kept PP lanes busy; here every lane has a point.
template <typename T, bool EXT>
__global__ void __launch_bounds__(64) k_implsch4_fin(const DevTab<T>* __restrict__ tp, int kijs, int kijl, const T* __restrict__ fin,
                                                     T* __restrict__ ffa, T* __restrict__ intfa, double* __restrict__ w2n) {
  const DevTab<T>& tb = *tp;
  const int ij = kijs + blockIdx.x * 64 + threadIdx.x;
  if (ij >= kijl) return;
  const T* fr = fin + (size_t)ij * V4_NFIN;
  T c[NSC];
  c[C_AIRD] = fr[FIN_AIRD]; c[C_UFRIC] = fr[FIN_UFRIC]; c[C_Z0M] = fr[FIN_Z0M];
  c[C_MIJ] = m_min(m_max(fr[FIN_MIJ], T(1)), T(V4_NFRE));   // a table index: whatever the row holds, stay inside FR(1:NFRE)
  c[C_ZPIFRMIJ] = tb.ZPIFR[(int)c[C_MIJ] - 1]; c[C_FR5MIJ] = tb.FR5[(int)c[C_MIJ] - 1];
  c[C_XS] = fr[FIN_XS]; c[C_YS] = fr[FIN_YS]; c[C_F1DCOS3] = fr[FIN_F1DCOS3]; c[C_F1DCOS2] = fr[FIN_F1DCOS2];
  c[C_F1DSIN2] = fr[FIN_F1DSIN2]; c[C_F1D] = fr[FIN_F1D]; c[C_RNFAC] = fr[FIN_RNFAC]; c[C_PHIWA] = fr[FIN_PHIWA];
  c[C_SINWD] = fr[FIN_SINWD]; c[C_COSWD] = fr[FIN_COSWD];
  const T AIRD = c[C_AIRD], UFRIC = c[C_UFRIC], sinwd = c[C_SINWD], coswd = c[C_COSWD];
  const T WSWAVE = fr[FIN_WSWAVE], CICOVER = fr[FIN_CICOVER], PHILF = fr[FIN_PHILF], XSTRESS = fr[FIN_XSTRESS], YSTRESS = fr[FIN_YSTRESS];
  stresso_point<T, EXT>(tb, c, true);
  const T PHIWA = c[C_PHIWA];
  T* fo = ffa + (size_t)ij * ECWAM_HIP_NFF;
  fo[8] = c[C_TAUW]; fo[9] = c[C_TAUWDIR];
  if (tb.LCFLX) {
    const T EPSUS3 = tb.EPSUS * m_sqrt(tb.EPSUS);
    // with an explicit ice attenuation term the blending with the ice-covered fluxes starts at CICOVER = 0 (wnfluxes.F90:206-214)
    const bool sdice_on = tb.LCIWA1 || tb.LCIWA2 || tb.LCIWA3;
    const T ZCITHRS = sdice_on ? T(0) : tb.CIBLOCK;
    const T CITHRSH_INV = sdice_on ? T(50) : T(1) / m_max(tb.CITHRSH, T(0.01));
    const T ZMAXEXP = sdice_on ? T(20) : T(10);
    T OOVAL = T(1), USTAR = UFRIC;
    T EM_OC = fr[FIN_EMEAN], F1_OC = fr[FIN_F1MEAN];
    if (tb.LICERUN && tb.LWAMRSETCI && CICOVER > ZCITHRS) {
      OOVAL = m_exp(-m_min(m_pow4(CICOVER * CITHRSH_INV), ZMAXEXP));
      const T U10P = m_max(WSWAVE, tb.EPSU10);
      const T CD_BULK = m_min((T(1.03E-3) + T(0.04E-3) * m_pow(U10P, T(1.48))) * m_pow(U10P, T(-0.21)), T(0.003));
      const T CD_WAVE = (UFRIC / U10P) * (UFRIC / U10P);
      const T CD_ICE = OOVAL * CD_WAVE + (T(1) - OOVAL) * CD_BULK;
      USTAR = m_max(m_sqrt(CD_ICE) * U10P, tb.EPSUS);
      if (tb.LWNEMOCOU) {  // fully developed sea under ice for the NEMO wave height / period (wnfluxes.F90:236-246)
        const T EFD_FAC = T(4) * tb.EGRCRV / (tb.G * tb.G);
        const T FFD_FAC = m_pow(tb.EGRCRV / tb.AFCRV, T(1) / tb.BFCRV) * tb.G;
        const T EFD = m_min(EFD_FAC * m_pow4(USTAR), T(6.25));
        EM_OC = m_max(OOVAL * EM_OC + (T(1) - OOVAL) * EFD, T(0.0625));
        const T FFD = FFD_FAC / USTAR;
        F1_OC = OOVAL * F1_OC + (T(1) - OOVAL) * FFD;
        F1_OC = m_min(m_max(F1_OC, tb.FR[1]), tb.FR[tb.NFRE - 1]);
      }
    }
    const T TAU = AIRD * m_max(USTAR * USTAR, tb.EPSUS);
    T TAUXD = TAU * sinwd, TAUYD = TAU * coswd;
    T TAUOCXD = TAUXD - OOVAL * XSTRESS, TAUOCYD = TAUYD - OOVAL * YSTRESS;
    const T TAUO = m_sqrt(TAUOCXD * TAUOCXD + TAUOCYD * TAUOCYD);
    const T TAUOC = m_min(m_max(TAUO / TAU, tb.TAUOCMIN), tb.TAUOCMAX);
    const T USTRA = fo[5], VSTRA = fo[6];
    if (tb.LWCOUAST && (USTRA != T(0) || VSTRA != T(0))) { TAUXD = USTRA; TAUOCXD = USTRA * TAUOC; TAUYD = VSTRA; TAUOCYD = VSTRA * TAUOC; }
    const T XN = AIRD * m_max(USTAR * USTAR * USTAR, EPSUS3);
    T PHIOCD = OOVAL * (PHILF - PHIWA) + (T(1) - OOVAL) * T(-3.75) * XN;
    const T PHIEPS = m_min(m_max(PHIOCD / XN, tb.PHIEPSMIN), tb.PHIEPSMAX);
    PHIOCD = PHIEPS * XN;
    const T PHIAW = OOVAL * PHIWA / XN + (T(1) - OOVAL) * T(3.75);
    T* io = intfa + (size_t)ij * ECWAM_HIP_NINTF;
    io[5] = TAUXD; io[6] = TAUYD; io[7] = TAUOCXD; io[8] = TAUOCYD; io[9] = TAUOC; io[10] = fr[FIN_TAUICX]; io[11] = fr[FIN_TAUICY];
    io[12] = PHIOCD; io[13] = PHIEPS; io[14] = PHIAW;
    if (tb.LWNEMOCOU && w2n) {  // wnfluxes.F90:304-328 (LNUPD = T; the ice stress of LWNEMOCOUWRS comes from the RARE build of the main kernel, zero otherwise)
      double* q = w2n + (size_t)ij * 13;
      q[3] = (double)PHIEPS; q[4] = (double)TAUOC;
      q[5] = (EM_OC != T(0)) ? 4.0 * (double)m_sqrt(EM_OC) : 0.0;
      q[6] = (F1_OC != T(0)) ? 1.0 / (double)F1_OC : 0.0;
      if (tb.LWNEMOTAUOC) { q[7] += (double)TAUOCXD; q[8] += (double)TAUOCYD; }
      else { q[7] += (double)TAUXD; q[8] += (double)TAUYD; }
      q[11] += (double)WSWAVE; q[12] += (double)PHIOCD;
      q[9] += (double)fr[FIN_TAUICX]; q[10] += (double)fr[FIN_TAUICY];      // (zero without LWNEMOCOUWRS)
    }
  }
  // stokestrn.F90:75-88 (LWNEMOCOUSTRN = F): the Stokes drift k_implsch4 left in INTFLDS
  if (tb.LWNEMOCOU && w2n && ((tb.LWNEMOCOUSEND && tb.LWCOU) || !tb.LWCOU)) {
    const T* io = intfa + (size_t)ij * ECWAM_HIP_NINTF;
    double* q = w2n + (size_t)ij * 13;
    q[0] = tb.LWNEMOCOUSTK ? (double)io[2] : 0.0;
    q[1] = tb.LWNEMOCOUSTK ? (double)io[3] : 0.0;
    if (tb.LWNEMOCOUSTRN) q[2] = (double)fr[FIN_STRNMS];      // cimsstrn.F90 (the RARE build of the main kernel)
  }
}
