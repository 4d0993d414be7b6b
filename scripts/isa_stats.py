"""Instruction histogram of one kernel in a gfx950 .s file: python scripts/isa_stats.py file.s kernel_substring

VALU instructions are split into the two issue classes measured on MI355X with scripts/valu_peak.hip: "fast" opcodes retire one
wave64 instruction per 2 cycles per SIMD (fma / mul / add / sub f32, mov, add / sub u32, right shifts, and / or / xor); every other
VALU opcode occupies a 4-cycle unit (min / max / med3, conversions, perm / bfe / left shifts / three-operand integer ops, compares,
cndmask; v_rcp-class 8), but only blocks issue for 2 cycles when fast instructions are interleaved.  Static lower bound per wave:
max(2 * VALU, 4 * slow) cycles."""
import collections
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32", "v_or_b32", "v_xor_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def base(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


s = open(sys.argv[1]).read()
pat = sys.argv[2]
# function bodies start at "<name>:" and end at s_endpgm
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = [l.split()[0] for l in body.splitlines()
           if l.strip() and not l.strip().startswith((';', '.', '//')) and not l.strip().endswith(':')]
    c = collections.Counter(ops)
    valu = {k: n for k, n in c.items() if k.startswith('v_')}
    fast = sum(n for k, n in valu.items() if base(k) in FAST)
    quarter = sum(n for k, n in valu.items() if base(k) in QUARTER)
    slow = sum(valu.values()) - fast - quarter
    print(name, 'total', sum(c.values()),
          'VALU', sum(valu.values()), f'(fast {fast}, slow {slow}, quarter-rate {quarter};',
          f'issue bound {2 * sum(valu.values())} cycles, slow-unit bound {4 * slow + 8 * quarter} cycles)',
          'SALU', sum(n for k, n in c.items() if k.startswith('s_')),
          'VMEM', sum(n for k, n in c.items() if k.startswith(('global', 'flat', 'buffer', 'scratch'))),
          'LDS', sum(n for k, n in c.items() if k.startswith('ds_')))
    print('  ' + ', '.join(f'{k}:{n}' for k, n in c.most_common(40)))
    break
