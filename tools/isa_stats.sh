#!/bin/bash
# dev helper (CPU box): register / spill figures of every k_frame instantiation, from the
# device assembly of frame.hip:  tools/isa_stats.sh [extra hipcc flags]  -> /tmp/frame_isa.s
# (what profiles/r06_head.md quotes: .sgpr_spill_count, .vgpr_spill_count, v_writelane / v_readlane)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fvisibility=hidden --cuda-device-only -S "$@" \
  -o /tmp/frame_isa.s meltingpot_amd/csrc/frame.hip || exit 1
python3 - <<'PY'
import re
txt = open("/tmp/frame_isa.s").read()
# per-kernel metadata blocks
for m in re.finditer(r"\.name:\s+(_ZN\S*k_frame\S*)\n(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))
    short = re.sub(r"^_ZN\d+_GLOBAL__N_\d+k_frameI", "", name)[:60]
    print(f"{short:62s} sgpr {get('sgpr_count'):3d} spill {get('sgpr_spill_count'):4d}  vgpr {get('vgpr_count'):3d} spill {get('vgpr_spill_count'):3d}")
# instruction mix of each function body
for m in re.finditer(r"^(_ZN\S*k_frame\S*):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
    body = m.group(2)
    short = re.sub(r"^_ZN\d+_GLOBAL__N_\d+k_frameI", "", m.group(1))[:60]
    n = len([l for l in body.splitlines() if re.match(r"\s+[sv]_|\s+(ds|global|buffer|flat)_", l)])
    print(f"{short:62s} {n:6d} instr  writelane {body.count('v_writelane'):4d} readlane {body.count('v_readlane_b32'):4d} "
          f"s_load {len(re.findall(r's_load_', body)):4d} scratch {len(re.findall(r'scratch_', body)):3d}")
PY
