// traverse_device.h — closest-hit traversal of the compressed 8-wide BVH (bvh8.h) with a per-lane stack in LDS and
// the watertight ray/triangle test. This is the MI355X replacement for the DXR TraceRay call at
// resources/shaders/raytracing/PathTracing.hlsl:71 (flags 0, mask 0xFF, opaque two-sided triangles,
// tmin < t < tmax). Equal-t ties resolve to the lower global triangle id so the result is independent of
// traversal order (and therefore of the acceleration structure).
#pragma once
#include "bvh8.h"
#include "device_math.h"

#ifndef PT_MIX_FMA
#define PT_MIX_FMA 1   // 1 = plane bytes enter the box test as binary16 denormals through v_fma_mix_f32 (see child_tests)
#endif
#ifndef PT_BOX_AND_PERM
#define PT_BOX_AND_PERM 1   // 1 = even plane bytes through a full-rate v_and_b32, odd ones through v_perm_b32; 0 = two byte permutes
#endif
#ifndef PT_BOX_SDWA_MASK
#define PT_BOX_SDWA_MASK 1   // 1 = hit-mask assembly of the box test with SDWA byte-select shifts and a sign-of-FMA decision (see child_tests)
#endif
#ifndef PT_BOX_BITSEL
#define PT_BOX_BITSEL 0   // 1 = near / far plane selection with v_bitop3_b32 bit selects (full rate) instead of v_cndmask_b32: -27 issue cycles per node step, but 3 more live registers
                        // -> spills at the pinned 6 waves per SIMD: C2 extend -1 %, C5 +-0 (profiles/r03/ab_box_mask_bitsel.txt); off
#endif
#ifndef PT_NODE_STRIDE_U4
#define PT_NODE_STRIDE_U4 5   // device stride of a BVH node in 16-byte units (5 = packed 80-byte nodes)
#endif

namespace ptd {

struct TraceRay {
  float ox, oy, oz, tmin;
  float dx, dy, dz, tmax;
};

struct HitRec {
  float t, u, v;
  uint32_t prim;
};

__device__ __forceinline__ float sel3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

// Watertight ray/triangle test (Woop, Benthin, Wald 2013), fp32 with the fp64 fallback for zero edge functions.
// The operation order here is the contract shared with the CPU oracle: keep it literal, no fused ops.
struct TriTester {
  float ox, oy, oz;
  int kx, ky, kz;
  float Sx, Sy, Sz;
  __device__ __forceinline__ void init(const TraceRay& r) {
    ox = r.ox; oy = r.oy; oz = r.oz;
    float ax = fabsf(r.dx), ay = fabsf(r.dy), az = fabsf(r.dz);
    kz = 0; float m = ax;
    if (ay > m) { kz = 1; m = ay; }
    if (az > m) { kz = 2; }
    kx = kz + 1; if (kx == 3) kx = 0;
    ky = kx + 1; if (ky == 3) ky = 0;
    float dz = sel3(r.dx, r.dy, r.dz, kz);
    if (dz < 0.0f) { int t = kx; kx = ky; ky = t; }
    Sx = sel3(r.dx, r.dy, r.dz, kx) / dz;
    Sy = sel3(r.dx, r.dy, r.dz, ky) / dz;
    Sz = 1.0f / dz;
  }
  // A vertex translated to the ray origin and sheared into the ray's frame: (x, y) in the plane across the ray, kz the
  // component along its dominant axis.
  struct Sheared { float x, y, kz; };
  __device__ __forceinline__ Sheared shear(float vx, float vy, float vz) const {
    const float ax = vx - ox, ay = vy - oy, az = vz - oz;
    Sheared r;
    r.kz = sel3(ax, ay, az, kz);
    r.x = sel3(ax, ay, az, kx) - Sx * r.kz;
    r.y = sel3(ax, ay, az, ky) - Sy * r.kz;
    return r;
  }
  __device__ __forceinline__ void test_sheared(const Sheared& A, const Sheared& B, const Sheared& C, uint32_t prim, float tmin, HitRec& best) const {
    const float Ax = A.x, Ay = A.y, Bx = B.x, By = B.y, Cx = C.x, Cy = C.y;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    // fp64 fallback only when an edge function is exactly zero. The wave-level test keeps it a real (rarely taken)
    // branch: without it the compiler if-converts the fp64 arithmetic into every triangle test.
    const bool edge_zero = (U == 0.0f || V == 0.0f || W == 0.0f);
    if (__ballot(edge_zero) != 0ull && edge_zero) {
      double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
      U = (float)(CxBy - CyBx);
      double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
      V = (float)(AxCy - AyCx);
      double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
      W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return;
    const float det = U + V + W;
    if (det == 0.0f) return;
    const float Az = Sz * A.kz, Bz = Sz * B.kz, Cz = Sz * C.kz;
    const float T = U * Az + V * Bz + W * Cz;
    const float rcpDet = 1.0f / det;
    const float t = T * rcpDet;
    if (!(t > tmin)) return;
    if (t < best.t || (t == best.t && best.prim != 0xFFFFFFFFu && prim < best.prim)) {  // t == tmax itself is not a hit
      best.t = t;
      best.u = V * rcpDet;
      best.v = W * rcpDet;
      best.prim = prim;
    }
  }
  __device__ __forceinline__ static Sheared pick(const Sheared& a, const Sheared& b, const Sheared& c, const Sheared& d, uint32_t i) {
    const bool odd = (i & 1u) != 0u, high = (i & 2u) != 0u;
    Sheared r;
    r.x = high ? (odd ? d.x : c.x) : (odd ? b.x : a.x);
    r.y = high ? (odd ? d.y : c.y) : (odd ? b.y : a.y);
    r.kz = high ? (odd ? d.kz : c.kz) : (odd ? b.kz : a.kz);
    return r;
  }
  // One primitive record (bvh8.h): triangle 0 = (v0, v1, v2); triangle 1, when present, picks its three vertices in its
  // own order out of {v0, v1, v2, v3}. Each test is the stand-alone arithmetic of that triangle. Returns triangles tested.
  __device__ __forceinline__ uint32_t test(const uint4& q0, const uint4& q1, const uint4& q2, const uint4& q3, float tmin, HitRec& best) const {
    const Sheared A = shear(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
    const Sheared B = shear(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z));
    const Sheared C = shear(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
    test_sheared(A, B, C, q0.w, tmin, best);
    if (q1.w == 0xFFFFFFFFu) return 1u;
    const Sheared D = shear(__uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z));
    const uint32_t code = q2.w;
    test_sheared(pick(A, B, C, D, code & 3u), pick(A, B, C, D, (code >> 2) & 3u), pick(A, B, C, D, (code >> 4) & 3u), q1.w, tmin, best);
    return 2u;
  }
};

__device__ __forceinline__ float safe_dir(float d) { return fabsf(d) < 1e-25f ? copysignf(1e-25f, d) : d; }

// Per-lane stack of 8-byte entries: the first `lds_entries` live in LDS ([entry][lane]: lane-contiguous, conflict-free
// ds_read/write_b64), deeper entries spill to a per-wave global region with the same layout. Typical traversals stay
// within the LDS part; the spill only bounds the worst case (2 entries per BVH level).
// The two halves are addressed through pointers that carry their address space (LDS = 3, global = 1): with generic
// pointers the compiler joins the branches into FLAT loads / stores, and a FLAT access to LDS still travels through the
// vector-memory pipe (the texture addresser), the kernel's scarcest resource.
typedef unsigned int pt_u2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) pt_u2v lds_u2;
typedef unsigned int pt_u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pt_u4v lds_u4;
typedef __attribute__((address_space(1))) pt_u2v glb_u2;
struct LaneStack {
  lds_u2* lds;       // this lane's slot of LDS entry 0
  glb_u2* spill;     // this lane's slot of spill entry 0
  int lds_entries;
  __device__ __forceinline__ void put(int sp, uint2 v) const {
    pt_u2v w;
    w.x = v.x; w.y = v.y;
    if (sp < lds_entries) lds[sp * 64] = w;
    else spill[(sp - lds_entries) * 64] = w;
  }
  __device__ __forceinline__ uint2 get(int sp) const {
    pt_u2v w;
    if (sp < lds_entries) w = lds[sp * 64];
    else w = spill[(sp - lds_entries) * 64];
    return make_uint2(w.x, w.y);
  }
};
// The same stack with its addresses rebuilt at every access from wave-uniform bases (scalar registers) and an opaque copy
// of the lane id, instead of three per-lane address registers that live through the whole kernel: two or three more
// instructions per push / pop, three registers fewer at the kernel's pressure peak (the box tests). For the instantiation
// that needs them (pair records + shared triangle phase: with LaneStack it reloads spilled addresses in every node step).
struct LaneStackCompact {
  lds_u2* lds_wave;     // wave-uniform: lane 0's slot of LDS entry 0
  glb_u2* spill_wave;   // wave-uniform: lane 0's slot of spill entry 0
  uint32_t lane;
  int lds_entries;
  __device__ __forceinline__ uint32_t lane_here() const {
    uint32_t l = lane;
    asm volatile("" : "+v"(l));   // opaque: the address arithmetic stays at the access
    return l;
  }
  __device__ __forceinline__ void put(int sp, uint2 v) const {
    pt_u2v w;
    w.x = v.x; w.y = v.y;
    const uint32_t l = lane_here();
    if (sp < lds_entries) lds_wave[(uint32_t)sp * 64u + l] = w;
    else spill_wave[(uint32_t)(sp - lds_entries) * 64u + l] = w;
  }
  __device__ __forceinline__ uint2 get(int sp) const {
    pt_u2v w;
    const uint32_t l = lane_here();
    if (sp < lds_entries) w = lds_wave[(uint32_t)sp * 64u + l];
    else w = spill_wave[(uint32_t)(sp - lds_entries) * 64u + l];
    return make_uint2(w.x, w.y);
  }
};

// Per-lane traversal state of one ray. The wave-level loop that drives it (replenishing finished lanes, postponing
// sparse triangle work) lives in k_extend; this struct holds what one lane carries between loop iterations.
struct Traverser {
  TraceRay ray;
  TriTester tt;
  HitRec best;
  float idx, idy, idz;
  uint32_t oct4;      // the octant (bit a set <=> direction component a is negative) replicated into the four bytes of a dword
  __device__ __forceinline__ uint32_t oct() const { return oct4 & 7u; }   // one full-rate AND where it is needed, not a register held through the kernel
  uint2 ng;           // pending node group: x = child_base, y = hit bits (top 8, by octant priority) | imask
  uint2 tg;           // pending triangle group: x = tri_base, y = triangle bits (low 24)
  int sp;
  bool risky;         // the fast slab form could overflow for this ray (see node_step)

  __device__ __forceinline__ void init(const TraceRay& r, float scene_mag) {
    ray = r;
    best.t = r.tmax; best.u = 0.0f; best.v = 0.0f; best.prim = 0xFFFFFFFFu;
    tt.init(r);
    const float sdx = safe_dir(r.dx), sdy = safe_dir(r.dy), sdz = safe_dir(r.dz);
    idx = 1.0f / sdx; idy = 1.0f / sdy; idz = 1.0f / sdz;
    oct4 = ((sdx < 0.0f ? 1u : 0u) | (sdy < 0.0f ? 2u : 0u) | (sdz < 0.0f ? 4u : 0u)) * 0x01010101u;
    ng = make_uint2(0u, 0x80000000u);  // root: one pending internal "child" at priority 7 of a group with base 0
    tg = make_uint2(0u, 0u);
    sp = 0;
    const float reach = fmaxf(scene_mag, fmaxf(fabsf(r.ox), fmaxf(fabsf(r.oy), fabsf(r.oz))));
    // 1e36 / 2^24 with PT_MIX_FMA: the fast form's plane coefficient carries an extra factor 2^24
    risky = !(reach * fmaxf(fabsf(idx), fmaxf(fabsf(idy), fabsf(idz))) < (PT_MIX_FMA ? 5e28f : 1e36f));
  }

  __device__ __forceinline__ bool has_node_work() const { return ng.y > 0x00FFFFFFu; }

  // Slab tests of the 8 quantised child boxes of one node. SAFE: (ax,bx) = (2^e, p - o), scaled by 1/d afterwards;
  // otherwise (ax,bx) = (2^e / d, (p - o) / d).
  template <bool SAFE>
  __device__ __forceinline__ void child_tests(const uint4& n1, const uint4& n2, const uint4& n3, const uint4& n4, float ax, float ay,
                                              float az, float bx, float by, float bz, bool negx, bool negy, bool negz,
                                              uint32_t oct4, uint32_t& hitmask) const {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t meta4 = h ? n1.w : n1.z;
      const uint32_t qlx = h ? n2.y : n2.x, qly = h ? n2.w : n2.z, qlz = h ? n3.y : n3.x;
      const uint32_t qhx = h ? n3.w : n3.z, qhy = h ? n4.y : n4.x, qhz = h ? n4.w : n4.z;
#if PT_BOX_BITSEL
      // near / far plane dwords by direction sign as full-rate bit selects against an all-ones / all-zeros mask (the sign of
      // the reciprocal direction spread over the word) instead of half-rate v_cndmask_b32; pinned as asm, the compiler turns
      // the C form back into compare + select
      uint32_t mx, my, mz, nx, fx, ny, fy, nz, fz;
      asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(mx) : "v"(idx));
      asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(my) : "v"(idy));
      asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(mz) : "v"(idz));
#define PT_BITSEL(dst, a, b, m) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(dst) : "v"(a), "v"(b), "v"(m))   /* m ? a : b, bitwise */
      PT_BITSEL(nx, qhx, qlx, mx); PT_BITSEL(fx, qlx, qhx, mx);
      PT_BITSEL(ny, qhy, qly, my); PT_BITSEL(fy, qly, qhy, my);
      PT_BITSEL(nz, qhz, qlz, mz); PT_BITSEL(fz, qlz, qhz, mz);
#undef PT_BITSEL
      (void)negx; (void)negy; (void)negz;
#else
      const uint32_t nx = negx ? qhx : qlx, fx = negx ? qlx : qhx;
      const uint32_t ny = negy ? qhy : qly, fy = negy ? qly : qhy;
      const uint32_t nz = negz ? qhz : qlz, fz = negz ? qlz : qhz;
#endif
      const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
      // 0b111 in every byte of an internal child (only the three octant bits are ever masked). Two shift-or instructions,
      // pinned as asm: the compiler folds every C form of this back into a quarter-rate v_mul_lo_u32.
      const uint32_t inner1 = is_inner4 >> 4;
      uint32_t inner3, inner_mask4;
      asm("v_lshl_or_b32 %0, %1, 1, %1" : "=v"(inner3) : "v"(inner1));
      asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(inner_mask4) : "v"(inner1), "v"(inner3));
      const uint32_t bit_index4 = (meta4 ^ (oct4 & inner_mask4)) & 0x1f1f1f1fu;
      const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
#if PT_MIX_FMA
      // Fast form only: the plane bytes go into the FMA as binary16 DENORMALS (a byte q in a 16-bit lane is the half
      // q * 2^-24; v_fma_mix_f32 widens it exactly), against coefficients pre-scaled by 2^24: q*2^-24 * (a*2^24) + b is
      // the same real number as q*a + b, rounded once either way, so the bits equal the convert-then-FMA form while two
      // planes share one byte-permute instead of one conversion each.
      // Bytes 0 and 2 of a plane dword become the two halves of one register with a single full-rate AND, bytes 1 and 3 with
      // one byte permute (tools/valurate: v_and_b32 issues in 2 cycles per wave, v_perm_b32 like most other VALU
      // instructions in 4): h[0] = children (0, 2) of this half, h[1] = children (1, 3).
      typedef _Float16 pt_h2 __attribute__((ext_vector_type(2)));
      pt_h2 hnx[2], hny[2], hnz[2], hfx[2], hfy[2], hfz[2];
      if (!SAFE) {
#define PT_H2_EVEN(q) __builtin_bit_cast(pt_h2, PT_BOX_AND_PERM ? ((q) & 0x00ff00ffu) : __builtin_amdgcn_perm(0u, (q), 0x0c020c00u))
#define PT_H2_ODD(q) __builtin_bit_cast(pt_h2, __builtin_amdgcn_perm(0u, (q), 0x0c030c01u))
        hnx[0] = PT_H2_EVEN(nx); hnx[1] = PT_H2_ODD(nx);
        hny[0] = PT_H2_EVEN(ny); hny[1] = PT_H2_ODD(ny);
        hnz[0] = PT_H2_EVEN(nz); hnz[1] = PT_H2_ODD(nz);
        hfx[0] = PT_H2_EVEN(fx); hfx[1] = PT_H2_ODD(fx);
        hfy[0] = PT_H2_EVEN(fy); hfy[1] = PT_H2_ODD(fy);
        hfz[0] = PT_H2_EVEN(fz); hfz[1] = PT_H2_ODD(fz);
#undef PT_H2_EVEN
#undef PT_H2_ODD
      }
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float tnx, tny, tnz, tfx, tfy, tfz;
#if PT_MIX_FMA
        if (!SAFE) {
          tnx = __builtin_fmaf((float)hnx[j & 1][j >> 1], ax, bx);
          tny = __builtin_fmaf((float)hny[j & 1][j >> 1], ay, by);
          tnz = __builtin_fmaf((float)hnz[j & 1][j >> 1], az, bz);
          tfx = __builtin_fmaf((float)hfx[j & 1][j >> 1], ax, bx);
          tfy = __builtin_fmaf((float)hfy[j & 1][j >> 1], ay, by);
          tfz = __builtin_fmaf((float)hfz[j & 1][j >> 1], az, bz);
        } else
#endif
        {
          tnx = __builtin_fmaf((float)((nx >> (8 * j)) & 0xffu), ax, bx);
          tny = __builtin_fmaf((float)((ny >> (8 * j)) & 0xffu), ay, by);
          tnz = __builtin_fmaf((float)((nz >> (8 * j)) & 0xffu), az, bz);
          tfx = __builtin_fmaf((float)((fx >> (8 * j)) & 0xffu), ax, bx);
          tfy = __builtin_fmaf((float)((fy >> (8 * j)) & 0xffu), ay, by);
          tfz = __builtin_fmaf((float)((fz >> (8 * j)) & 0xffu), az, bz);
        }
        if (SAFE) { tnx *= idx; tny *= idy; tnz *= idz; tfx *= idx; tfy *= idy; tfz *= idz; }
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, ray.tmin));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
        // conservative: widen by ~8 ulp so fp32 rounding of the slab arithmetic can never cull a true hit
#if PT_BOX_SDWA_MASK
        // Mask assembly at issue cost (tools/valurate: shifts, bit-field extracts, compares and selects are half-rate; FMA
        // and the bit-logic forms full-rate). (1) The child's mask bits cb << bi with ONE sub-dword-addressed shift that
        // reads byte j of both packed registers, instead of two byte extractions and a shift. (2) The decision
        // tn <= tf * 1.000001 as the sign of ONE fused multiply-add tf * 1.000001 - tn (exact sign of the real difference:
        // +0 on equality, a hit; tn and tf are never NaN, fmaxf / fminf against the finite tmin / best.t drop them),
        // spread over the word by an arithmetic shift and applied with and-not + or, instead of multiply, compare,
        // select and or. The compare now sees the exact product rather than the rounded one: half an ulp of the 8-ulp
        // margin, and tests/walker mirrors it so the visit counts stay equal to the kernel's.
        uint32_t m;
        switch (j) {
          case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(m) : "v"(bit_index4), "v"(child_bits4)); break;
          case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(m) : "v"(bit_index4), "v"(child_bits4)); break;
          case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(m) : "v"(bit_index4), "v"(child_bits4)); break;
          default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(m) : "v"(bit_index4), "v"(child_bits4)); break;
        }
        const float dlt = __builtin_fmaf(tf, 1.000001f, -tn);
        uint32_t miss;   // all ones: the interval is empty. Both pinned as asm: the compiler rewrites the C forms into compare + select
        asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(miss) : "v"(dlt));
        asm("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xba" : "+v"(hitmask) : "v"(m), "v"(miss));   // hitmask | (m & ~miss)
#else
        if (tn <= tf * 1.000001f) {
          const uint32_t cb = (child_bits4 >> (8 * j)) & 0xffu;
          const uint32_t bi = (bit_index4 >> (8 * j)) & 0xffu;
          hitmask |= cb << bi;
        }
#endif
      }
    }
  }

  // Pops the nearest pending internal child, fetches its 80-byte node (5 x 128-bit loads) and intersects the 8
  // quantised child boxes.
  // wave_risky (wave-uniform): some lane of the wave holds a ray flagged `risky` (refreshed at every refill).
  // top / top_nodes: the first top_nodes nodes of the array (its breadth-first top levels) also sit in LDS; lanes whose
  // node is among them read it there and stay off the vector-memory pipe.
  template <class STACK>
  __device__ __forceinline__ void node_step(const uint4* __restrict__ nodes, const STACK& stk, bool wave_risky,
                                            const lds_u4* top, uint32_t top_nodes) {
    const uint32_t hits = ng.y;
    const int b = 31 - __clz((int)hits);
    ng.y &= ~(1u << b);
    if (ng.y > 0x00FFFFFFu) { stk.put(sp, ng); ++sp; }
    const uint32_t slot = ((uint32_t)b - 24u) ^ oct();
    const uint32_t rel = __popc(hits & ~(0xFFFFFFFFu << slot));
    // byte offset in 32 bits (pt_scene_create rejects BVHs beyond 4 GB of nodes): x80 as shift-adds, base stays scalar
    const uint32_t ni = ng.x + rel;
#if PT_NODE_STRIDE_U4 == 5
    uint32_t noff;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(noff) : "v"(ni));   // x5 (asm: the compiler turns shift-adds into v_mul_lo_u32)
    noff <<= 4;
#else
    const uint32_t noff = ni * (PT_NODE_STRIDE_U4 * 16u);
#endif
    const uint4* np = (const uint4*)((const char*)nodes + noff);
    uint4 n0, n1, n2, n3, n4;
    if (ni < top_nodes) {
      const lds_u4* lp = (const lds_u4*)((const __attribute__((address_space(3))) char*)top + noff);
      const pt_u4v a0 = lp[0], a1 = lp[1], a2 = lp[2], a3 = lp[3], a4 = lp[4];
      n0 = make_uint4(a0.x, a0.y, a0.z, a0.w); n1 = make_uint4(a1.x, a1.y, a1.z, a1.w); n2 = make_uint4(a2.x, a2.y, a2.z, a2.w);
      n3 = make_uint4(a3.x, a3.y, a3.z, a3.w); n4 = make_uint4(a4.x, a4.y, a4.z, a4.w);
    } else {
      n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3]; n4 = np[4];
    }

    const uint32_t e = n0.w;
#if PT_MIX_FMA
    // 2^e * 2^24, built in the exponent field: byte -> (byte << 23) + (24 << 23), one v_lshl_add_u32 per axis (pinned as
    // asm like the other shift-adds). e + 24 < 255 for every scene the fast form is used on: coordinates beyond 2^103 make
    // every ray risky, and the safe form takes the bias off again.
    uint32_t sx24u, sy24u, sz24u;
    {
      const uint32_t bias = 0x0C000000u, ex = e & 0xffu, ey = (e >> 8) & 0xffu, ez = (e >> 16) & 0xffu;
      asm("v_lshl_add_u32 %0, %1, 23, %2" : "=v"(sx24u) : "v"(ex), "s"(bias));
      asm("v_lshl_add_u32 %0, %1, 23, %2" : "=v"(sy24u) : "v"(ey), "s"(bias));
      asm("v_lshl_add_u32 %0, %1, 23, %2" : "=v"(sz24u) : "v"(ez), "s"(bias));
    }
#else
    const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23),
                sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
#endif
    const float px = __uint_as_float(n0.x) - ray.ox, py = __uint_as_float(n0.y) - ray.oy, pz = __uint_as_float(n0.z) - ray.oz;
    // the octant bits, read off the reciprocal directions (same signs by construction in init: one compare each)
    const bool negx = idx < 0.0f, negy = idy < 0.0f, negz = idz < 0.0f;
    uint32_t hitmask = 0u;
    // Fast form: t = q * (2^e / d) + (p - o) / d, one FMA per plane. Its two terms can overflow fp32 although their sum
    // is small (|scene coordinate| / |d| > ~1e37: astronomically large scenes, or near-axis-parallel rays in large
    // ones); such rays (flagged once, at init) take the safe form t = (q * 2^e + (p - o)) / d, where only a genuinely
    // huge distance overflows, with the right sign. The wave-level test keeps the safe form out of ordinary waves.
    if (wave_risky && risky) {
#if PT_MIX_FMA
      const float sx = __uint_as_float(sx24u - 0x0C000000u), sy = __uint_as_float(sy24u - 0x0C000000u), sz = __uint_as_float(sz24u - 0x0C000000u);
#endif
      child_tests<true>(n1, n2, n3, n4, sx, sy, sz, px, py, pz, negx, negy, negz, oct4, hitmask);
    } else {
#if PT_MIX_FMA
      child_tests<false>(n1, n2, n3, n4, __uint_as_float(sx24u) * idx, __uint_as_float(sy24u) * idy, __uint_as_float(sz24u) * idz, px * idx, py * idy, pz * idz,
                         negx, negy, negz, oct4, hitmask);
#else
      child_tests<false>(n1, n2, n3, n4, sx * idx, sy * idy, sz * idz, px * idx, py * idy, pz * idz, negx, negy, negz, oct4, hitmask);
#endif
    }
    ng.x = n1.x;
    ng.y = (hitmask & 0xFF000000u) | (e >> 24);
    tg.x = n1.y;
    tg.y = hitmask & 0x00FFFFFFu;
  }

  // Tests the lowest pending record of tg. PAIRS: 64-byte records (one triangle, or two that share an edge); else 48-byte
  // single-triangle records. Returns triangles tested.
  template <bool PAIRS>
  __device__ __forceinline__ uint32_t tri_step(const uint4* __restrict__ tris) {
    const int i = __ffs((int)tg.y) - 1;
    tg.y &= tg.y - 1u;
    const uint32_t ti = tg.x + (uint32_t)i;
    if (PAIRS) {
      const uint4* tp = (const uint4*)((const char*)tris + (ti << 6));   // 32-bit byte offset (< 4 GB of records)
      const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
      return tt.test(q0, q1, q2, q3, ray.tmin, best);
    }
    uint32_t toff;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(toff) : "v"(ti));   // x3, then x16 (asm: the compiler turns shift-adds into v_mul_lo_u32)
    const uint4* tp = (const uint4*)((const char*)tris + (toff << 4));
    const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
    tt.test_sheared(tt.shear(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)),
                    tt.shear(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                    tt.shear(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z)), q0.w, ray.tmin, best);
    return 1u;
  }
};

}  // namespace ptd
