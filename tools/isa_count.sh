#!/bin/bash
# VALU / v_bitop3 / scratch instruction counts and register use per kernel of one .hip file (device ISA only).
# usage: tools/isa_count.sh binius_amd/csrc/kernels_mul9.hip [extra hipcc flags]
f=$1; shift
o=$(mktemp -d)/out.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S "$f" -o "$o" "$@" || exit 1
python3 - "$o" <<'PY'
import sys, re
cur = None; cnt = {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1); cnt[cur] = dict(valu=0, bitop3=0, scratch=0); continue
    if cur is None: continue
    t = l.strip().split()
    if not t: continue
    op = t[0]
    if op.startswith('v_'): cnt[cur]['valu'] += 1
    if op.startswith('v_bitop3'): cnt[cur]['bitop3'] += 1
    if op.startswith('scratch_'): cnt[cur]['scratch'] += 1
    m = re.match(r'\.(vgpr_count|vgpr_spill_count):\s*(\d+)', l.strip())
    if m: pass
for k, v in cnt.items():
    if v['valu'] > 50: print(k[:80], v)
name = None
for l in open(sys.argv[1]):
    s = l.strip()
    if s.startswith('.name:') and '_Z' in s: name = s.split()[-1][:60]
    if s.startswith('.vgpr_count:') or s.startswith('.vgpr_spill_count:') or s.startswith('.agpr_count:'):
        print(' ', name, s)
PY
