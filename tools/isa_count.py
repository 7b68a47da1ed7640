"""Static instruction mix per function of an AMDGPU .s file (used to find VALU-heavy helpers)."""
import re, sys
from collections import Counter
cur = None; counts = {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', l)
    if m: cur = m.group(1); counts[cur] = Counter(); continue
    if re.match(r'^\.Lfunc_end', l): cur = None; continue
    t = l.strip()
    if cur and t and not t.startswith((';', '.')) and not t.endswith(':'):
        op = t.split()[0]
        k = 'valu' if op.startswith('v_') else 'smem' if op.startswith('s_load') else 'salu' if op.startswith('s_') else 'vmem' if ('load' in op or 'store' in op) else 'other'
        counts[cur][k] += 1
        if op in ('v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_div_scale_f32'): counts[cur][op] += 1
for k, v in counts.items(): print(k[:60], dict(v))
