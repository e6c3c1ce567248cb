#!/usr/bin/env python3
"""Hazard check of the K1 instances' ISA: the sorting networks are hand-scheduled inline asm (bare v_min / v_max,
v_pk_min/max_i16 under rewritten EXEC masks, hand-counted s_nop) followed by compiler-generated DPP moves.  On gfx9-family
hardware (gfx950 included) the program must keep
    VALU writes a VGPR  ->  a DPP instruction reads it as its DPP source:   2 wait states
    a VALU instruction writes EXEC (v_cmpx*)  ->  a DPP instruction:        5 wait states
      (EXEC written by the scalar unit — s_mov / s_and / s_or ... exec — needs none: LLVM's GCNHazardRecognizer::checkDPPHazards)
(one wait state per intervening instruction, s_nop N = N + 1).  The assembler's hazard recognizer handles what the compiler
emits; inline asm is opaque text to the scheduler, so this script re-checks the final ISA of a translation unit:

    python3 tools/check_dpp_hazards.py <dtype 0|1> <all 0|1>      # compiles rank_stats_inst.hip device-only to ISA, exit 1 on a violation
    python3 tools/check_dpp_hazards.py file.s                      # checks an existing listing
"""
import os
import re
import subprocess
import sys
import tempfile

DPP_MOD = re.compile(r'\b(quad_perm:|row_shl:|row_shr:|row_ror:|row_mirror|row_half_mirror|row_bcast:|row_newbcast:|row_share:|row_xmask:|wave_shl|wave_shr|wave_rol|wave_ror)')
REG = re.compile(r'^v(\d+)$|^v\[(\d+):(\d+)\]$')


def regs(tok):
    m = REG.match(tok.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def parse(line):
    line = line.split(';')[0].split('//')[0].strip()
    if not line or line.endswith(':') or line.startswith('.'):
        return None
    parts = line.split(None, 1)
    mnem = parts[0]
    ops = [o.strip() for o in re.split(r',\s*', parts[1])] if len(parts) > 1 else []
    return mnem, ops, line


def check(text):
    """-> list of (function, line number, message)"""
    bad = []
    func = '?'
    window = []                 # (wait states since, mnem, dst regs, writes exec, text)
    for ln, raw in enumerate(text.splitlines(), 1):
        s = raw.strip()
        m = re.match(r'^([A-Za-z_][\w$.]*):', s)
        if m and not s.startswith('.L') and not s.startswith(';'):
            func = m.group(1)
            window = []
            continue
        p = parse(raw)
        if p is None:
            continue
        mnem, ops, txt = p
        is_dpp = mnem.startswith('v_') and (mnem.endswith('_dpp') or DPP_MOD.search(txt) is not None)
        if is_dpp and len(ops) >= 2:
            # the DPP source is src0: the first operand after the destination (its last token may carry the modifiers)
            src0 = ops[1].split()[0]
            need = regs(src0)
            dist = 0
            for mn, dst, wexec, t in reversed(window):
                if mn.startswith('v_') and need & dst and dist < 2:
                    bad.append((func, ln, 'DPP reads %s %d wait state(s) after VALU write: "%s" -> "%s"' % (src0, dist, t, txt)))
                if wexec and dist < 5:
                    bad.append((func, ln, 'DPP %d wait state(s) after an EXEC write: "%s" -> "%s"' % (dist, t, txt)))
                dist += wait_states(mn, t)
                if dist >= 5:
                    break
        dst = regs(ops[0].split()[0]) if (ops and mnem.startswith('v_')) else set()
        wexec = mnem.startswith('v_cmpx') or (mnem.startswith('v_') and bool(ops) and ops[0].split()[0] in ('exec', 'exec_lo', 'exec_hi'))
        window.append((mnem, dst, wexec, txt))
        if len(window) > 8:
            window.pop(0)
    return bad


def wait_states(mnem, txt):
    if mnem == 's_nop':
        try:
            return int(txt.split()[1], 0) + 1
        except Exception:
            return 1
    return 1


def compile_isa(dtype, all_tests, extra=()):
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nanomod_amd', 'csrc', 'rank_stats_inst.hip')
    out = os.path.join(tempfile.mkdtemp(), 'k1_d%s_a%s.s' % (dtype, all_tests))
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', '-DNMOD_INST_DTYPE=%s' % dtype,
           '-DNMOD_INST_ALL=%s' % all_tests, '--offload-device-only', '-S', src, '-o', out] + list(extra)
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def main(argv):
    if len(argv) == 2 and os.path.exists(argv[1]):
        path = argv[1]
    else:
        path = compile_isa(argv[1], argv[2], argv[3:])
    text = open(path).read()
    n_dpp = sum(1 for ln in text.splitlines() if DPP_MOD.search(ln.split(';')[0]))
    bad = check(text)
    for f, ln, msg in bad[:40]:
        print('%s:%d: [%s] %s' % (path, ln, f[-60:], msg))
    print('%s: %d DPP instructions, %d hazard violations' % (path, n_dpp, len(bad)))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
