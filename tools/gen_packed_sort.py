#!/usr/bin/env python3
"""Generator of nanomod_amd/csrc/packed_sort_i16.hpp: the sorting network of the int16 K1 kernels on PACKED keys.

A lane holds 16 int16 keys in 8 VGPRs (two per register).  v_pk_min_i16 / v_pk_max_i16 compare-exchange two pairs of
keys per instruction pair — twice the keys per instruction of the float network — provided the two compare-exchanges of
an instruction pair take one key from register A and one from register B each.  The generator walks the bitonic
network (mirror formulation, ascending everywhere), tracks which (register, half) every wire of the network lives in,
pairs the compare-exchanges of every stage by registers and emits

  * register-register stages: v_pk_min_i16 + v_pk_max_i16 (op_sel swaps the halves of B where the partner sits in the
    other half); the minima land in one register, the maxima in the other — the wire map is updated, nothing moves;
  * the stage whose partners share a register (one per in-lane merge): v_min_i16 / v_max_i16 with SDWA half selects,
    two instructions per register;
  * cross-lane stages (whole registers travel: both halves belong to the same lane role): one DPP move per register,
    then v_pk_min_i16 under EXEC = the lanes that keep the minimum and v_pk_max_i16 under the complement — three
    instructions per two keys where the float network spends four.

Run: python3 tools/gen_packed_sort.py   (self-tests the emitted network on random keys before writing the header)."""
import os
import random
import sys

R = 16          # keys per lane
NREG = 8


class Net:
    def __init__(self, LG):
        self.LG = LG
        self.regs = [[j, j + 8] for j in range(NREG)]        # register j: wires (lo, hi); c0 = 8
        self.ops = []                                         # the emitted program

    def loc(self, w):
        for j, (a, b) in enumerate(self.regs):
            if a == w:
                return j, 0
            if b == w:
                return j, 1
        raise KeyError(w)

    # ---- in-lane stage: partner(w), lower(w) -> bool (the wire that keeps the minimum)
    def in_lane(self, partner, lower):
        done = set()
        for ia in range(NREG):
            if ia in done:
                continue
            a1, a2 = self.regs[ia]
            ib, h1 = self.loc(partner(a1))
            if ib == ia:                                      # the two wires of a register are partners
                assert partner(a1) == a2
                lo_w, hi_w = (a1, a2) if lower(a1) else (a2, a1)
                self.ops.append(('within', ia))
                self.regs[ia] = [lo_w, hi_w]
                done.add(ia)
                continue
            ib2, h2 = self.loc(partner(a2))
            assert ib2 == ib and h2 == 1 - h1, 'the stage does not pair registers'
            swapped = h1 == 1
            b1, b2 = partner(a1), partner(a2)
            low1, up1 = (a1, b1) if lower(a1) else (b1, a1)
            low2, up2 = (a2, b2) if lower(a2) else (b2, a2)
            self.ops.append(('pk', ia, ib, swapped))
            self.regs[ia] = [low1, low2]
            self.regs[ib] = [up1, up2]
            done.update((ia, ib))

    def mirror_in_lane(self, size):                            # blocks of `size` wires: w <-> w ^ (size - 1)
        self.in_lane(lambda w: w ^ (size - 1), lambda w: (w & (size >> 1)) == 0)

    def hc_in_lane(self, d):
        self.in_lane(lambda w: w ^ d, lambda w: (w & d) == 0)

    # ---- cross-lane stages
    def hc_lanes(self, D):
        self.ops.append(('xhc', D))

    def mirror_lanes(self, G):
        src = []
        for j in range(NREG):
            lo, hi = self.regs[j]
            jj, h = self.loc(R - 1 - lo)
            jj2, h2 = self.loc(R - 1 - hi)
            assert jj == jj2 and h2 == 1 - h
            src.append((jj, h == 1))                          # partner register, halves swapped?
        self.ops.append(('xmir', G, src))

    def build(self):
        # 16 keys of a lane: bitonic sort, mirror formulation
        size = 2
        while size <= R:
            self.mirror_in_lane(size)
            d = size >> 2
            while d >= 1:
                self.hc_in_lane(d)
                d >>= 1
            size <<= 1
        G = 2
        while G <= self.LG:
            self.mirror_lanes(G)
            D = G >> 2
            while D >= 1:
                self.hc_lanes(D)
                D >>= 1
            d = R >> 1
            while d >= 1:
                self.hc_in_lane(d)
                d >>= 1
            G <<= 1
        return self

    # ---- simulation of the emitted program on [lane][register] = (lo, hi)
    def simulate(self, keys):
        LG = self.LG
        regs0 = [[j, j + 8] for j in range(NREG)]
        st = [[[keys[l][regs0[j][0]], keys[l][regs0[j][1]]] for j in range(NREG)] for l in range(LG)]
        for op in self.ops:
            if op[0] == 'pk':
                _, ia, ib, sw = op
                for l in range(LG):
                    A, B = st[l][ia], st[l][ib]
                    Bs = [B[1], B[0]] if sw else B
                    st[l][ia], st[l][ib] = [min(A[0], Bs[0]), min(A[1], Bs[1])], [max(A[0], Bs[0]), max(A[1], Bs[1])]
            elif op[0] == 'within':
                for l in range(LG):
                    A = st[l][op[1]]
                    st[l][op[1]] = [min(A), max(A)]
            elif op[0] == 'xhc':
                D = op[1]
                new = [[None] * NREG for _ in range(LG)]
                for l in range(LG):
                    for j in range(NREG):
                        A, T = st[l][j], st[l ^ D][j]
                        f = min if (l & D) == 0 else max
                        new[l][j] = [f(A[0], T[0]), f(A[1], T[1])]
                st = new
            else:
                _, G, src = op
                new = [[None] * NREG for _ in range(LG)]
                for l in range(LG):
                    for j in range(NREG):
                        jj, sw = src[j]
                        A, T = st[l][j], st[l ^ (G - 1)][jj]
                        Ts = [T[1], T[0]] if sw else T
                        f = min if (l & (G >> 1)) == 0 else max
                        new[l][j] = [f(A[0], Ts[0]), f(A[1], Ts[1])]
                st = new
        out = []
        for l in range(LG):
            row = [None] * R
            for j in range(NREG):
                row[self.regs[j][0]] = st[l][j][0]
                row[self.regs[j][1]] = st[l][j][1]
            out.extend(row)
        return out

    def cost(self):
        c = 0
        for op in self.ops:
            c += {'pk': 2, 'within': 2, 'xhc': 3 * NREG, 'xmir': 3 * NREG}[op[0]]
        return c


LANE_BIT_MASK = {0: 0x5555555555555555, 1: 0x3333333333333333, 2: 0x0f0f0f0f0f0f0f0f, 3: 0x00ff00ff00ff00ff,
                 4: 0x0000ffff0000ffff, 5: 0x00000000ffffffff}


def emit(net):
    LG = net.LG
    L = []
    L.append('template <>\n__device__ __forceinline__ void seg_sort_packed16<%d>(unsigned (&p)[8], int lane) {' % LG)
    L.append('  (void)lane;')
    L.append('  unsigned t0, t1, t2, t3, t4, t5, t6, t7;')
    for op in net.ops:
        if op[0] == 'pk':
            _, ia, ib, sw = op
            L.append('  pk_ce<%s>(p[%d], p[%d]);' % ('true' if sw else 'false', ia, ib))
        elif op[0] == 'within':
            L.append('  pk_ce_within(p[%d]);' % op[1])
        elif op[0] == 'xhc':
            D = op[1]
            bit = D.bit_length() - 1
            L.append('  pk_dpp_guard(p);')
            for j in range(NREG):
                L.append('  t%d = pk_lane_xor<%d>(p[%d]);' % (j, D, j))
            L.append('  pk_lane_stage<%s>(p, t0, t1, t2, t3, t4, t5, t6, t7, 0x%016xull);' % ('0', LANE_BIT_MASK[bit]))
        else:
            _, G, src = op
            bit = (G >> 1).bit_length() - 1
            sws = set(sw for _, sw in src)
            assert len(sws) == 1, 'mixed half orders in a mirror stage'
            L.append('  pk_dpp_guard(p);')
            for j in range(NREG):
                L.append('  t%d = pk_lane_mirror<%d>(p[%d], lane);' % (j, G, src[j][0]))
            L.append('  pk_lane_stage<%s>(p, t0, t1, t2, t3, t4, t5, t6, t7, 0x%016xull);' % ('1' if sws.pop() else '0', LANE_BIT_MASK[bit]))
    L.append('}')
    # the sorted keys as floats: wire w of the lane -> x[w]
    L.append('template <>\n__device__ __forceinline__ void unpack_sorted16<%d>(const unsigned (&p)[8], float (&x)[16]) {' % LG)
    for w in range(R):
        j, h = net.loc(w)
        L.append('  x[%d] = (float)(short)(p[%d]%s);' % (w, j, ' >> 16' if h else ' & 0xffffu'))
    L.append('}')
    return '\n'.join(L)


HEADER = '''// GENERATED by tools/gen_packed_sort.py — do not edit (the generator documents and self-tests the network).
//
// Sort of the LG x 16 int16 keys of every LG-lane group of a wave, two keys per VGPR: ascending by key index
// 16 * lane + w, w = the wire of the network; unpack_sorted16 hands them out as floats in wire order.
// VALU instructions per wave: %s (the float network of the same shape: 718).
#pragma once
#include <hip/hip_runtime.h>
#include "wave_ops.hpp"

namespace nmod {

// compare-exchange of the two key pairs (a.lo, b.[SW ? hi : lo]) and (a.hi, b.[SW ? lo : hi]): the minima into a, the maxima into b
template <bool SW>
__device__ __forceinline__ void pk_ce(unsigned& a, unsigned& b) {
  unsigned mn, mx;
  if constexpr (SW) {
    asm("v_pk_min_i16 %%0, %%2, %%3 op_sel:[0,1] op_sel_hi:[1,0]\\n\\tv_pk_max_i16 %%1, %%2, %%3 op_sel:[0,1] op_sel_hi:[1,0]"
        : "=&v"(mn), "=&v"(mx) : "v"(a), "v"(b));
  } else {
    asm("v_pk_min_i16 %%0, %%2, %%3\\n\\tv_pk_max_i16 %%1, %%2, %%3" : "=&v"(mn), "=&v"(mx) : "v"(a), "v"(b));
  }
  a = mn; b = mx;
}
// the two keys of one register: minimum into the low half, maximum into the high half
__device__ __forceinline__ void pk_ce_within(unsigned& a) {
  unsigned r;
  asm("v_min_i16_sdwa %%0, %%1, %%1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\\n\\t"
      "v_max_i16_sdwa %%0, %%1, %%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1"
      : "=&v"(r) : "v"(a));
  a = r;
}
// a VGPR written by VALU needs two wait states before a DPP instruction reads it; the compiler's hazard recognizer does not
// look into the inline asm above, so the distance is kept by hand in front of every group of DPP moves
// (the registers pass through the statement, so it stays between their writers and the moves)
__device__ __forceinline__ void pk_dpp_guard(unsigned (&p)[8]) {
  asm volatile("s_nop 1" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
}
template <int M>
__device__ __forceinline__ unsigned pk_lane_xor(unsigned v) { return __float_as_uint(lane_xor<M>(__uint_as_float(v))); }
template <int G>
__device__ __forceinline__ unsigned pk_lane_mirror(unsigned v, int lane) { return __float_as_uint(lane_mirror<G>(__uint_as_float(v), lane)); }
// one cross-lane stage on all eight registers: the lanes of `low_mask` keep the minima, the others the maxima
// (SW: the partner's halves are swapped — mirror stages)
template <int SW>
__device__ __forceinline__ void pk_lane_stage(unsigned (&p)[8], unsigned t0, unsigned t1, unsigned t2, unsigned t3, unsigned t4,
                                              unsigned t5, unsigned t6, unsigned t7, unsigned long long low_mask) {
#if defined(NMOD_PK_SELECT)
  // variant without EXEC writes: both results, then a select by the lane mask (4 instructions per register)
  const unsigned tt[8] = {t0, t1, t2, t3, t4, t5, t6, t7};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned mn, mx;
    if constexpr (SW) asm("v_pk_min_i16 %%0, %%2, %%3 op_sel:[0,1] op_sel_hi:[1,0]\\n\\tv_pk_max_i16 %%1, %%2, %%3 op_sel:[0,1] op_sel_hi:[1,0]\\n\\tv_cndmask_b32_e64 %%0, %%1, %%0, %%4"
                          : "=&v"(mn), "=&v"(mx) : "v"(p[j]), "v"(tt[j]), "s"(low_mask));
    else asm("v_pk_min_i16 %%0, %%2, %%3\\n\\tv_pk_max_i16 %%1, %%2, %%3\\n\\tv_cndmask_b32_e64 %%0, %%1, %%0, %%4"
             : "=&v"(mn), "=&v"(mx) : "v"(p[j]), "v"(tt[j]), "s"(low_mask));
    p[j] = mn;
  }
  return;
#endif
  unsigned long long save;
#define NMOD_PK8(OP, SEL) \\
  OP " %%0, %%0, %%9" SEL "\\n\\t" OP " %%1, %%1, %%10" SEL "\\n\\t" OP " %%2, %%2, %%11" SEL "\\n\\t" OP " %%3, %%3, %%12" SEL "\\n\\t" \\
  OP " %%4, %%4, %%13" SEL "\\n\\t" OP " %%5, %%5, %%14" SEL "\\n\\t" OP " %%6, %%6, %%15" SEL "\\n\\t" OP " %%7, %%7, %%16" SEL "\\n\\t"
  if constexpr (SW) {
    asm("s_mov_b64 %%8, exec\\n\\ts_and_b64 exec, %%8, %%17\\n\\t"
                 NMOD_PK8("v_pk_min_i16", " op_sel:[0,1] op_sel_hi:[1,0]")
                 "s_andn2_b64 exec, %%8, %%17\\n\\t"
                 NMOD_PK8("v_pk_max_i16", " op_sel:[0,1] op_sel_hi:[1,0]")
                 "s_mov_b64 exec, %%8\\n\\ts_nop 4"       // (EXEC write -> the next stage's DPP moves: 5 wait states, by hand)
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "=&s"(save)
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "s"(low_mask)
                 : "scc");            // (s_and_b64 / s_andn2_b64 write SCC: without the clobber a loop compare can be live across the block)
  } else {
    asm("s_mov_b64 %%8, exec\\n\\ts_and_b64 exec, %%8, %%17\\n\\t"
                 NMOD_PK8("v_pk_min_i16", "")
                 "s_andn2_b64 exec, %%8, %%17\\n\\t"
                 NMOD_PK8("v_pk_max_i16", "")
                 "s_mov_b64 exec, %%8\\n\\ts_nop 4"       // (EXEC write -> the next stage's DPP moves: 5 wait states, by hand)
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "=&s"(save)
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "s"(low_mask)
                 : "scc");            // (s_and_b64 / s_andn2_b64 write SCC: without the clobber a loop compare can be live across the block)
  }
#undef NMOD_PK8
}

template <int LG> __device__ __forceinline__ void seg_sort_packed16(unsigned (&p)[8], int lane);
template <int LG> __device__ __forceinline__ void unpack_sorted16(const unsigned (&p)[8], float (&x)[16]);

'''


def main():
    random.seed(1)
    bodies, costs = [], []
    for LG in (8, 16):
        net = Net(LG).build()
        for trial in range(300):
            mode = trial % 3
            keys = [[(random.randint(-32768, 32767) if mode == 0 else random.randint(-3, 3) if mode == 1 else
                      random.choice([32767, -32768, 0])) for _ in range(R)] for _ in range(LG)]
            out = net.simulate(keys)
            assert out == sorted(v for row in keys for v in row), (LG, trial)
        bodies.append(emit(net))
        costs.append('%d (LG = %d)' % (net.cost(), LG))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nanomod_amd', 'csrc', 'packed_sort_i16.hpp')
    with open(path, 'w') as f:
        f.write((HEADER % ', '.join(costs)).replace('%%', '%'))
        f.write('\n\n'.join(bodies))
        f.write('\n\n}  // namespace nmod\n')
    print('wrote', os.path.normpath(path), 'VALU per wave:', ', '.join(costs))


if __name__ == '__main__':
    sys.exit(main())
