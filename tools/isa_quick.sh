#!/bin/bash
# dev helper (CPU box): the device assembly of the draw-only and clean_up k_frame kernels alone (-DMP_FRAME_ISA_SUBSET:
# 20 s instead of 100) -> /tmp/frame_q.s, and the draw-only WORLD.RGB kernel's body -> /tmp/frame_q_nt1.s
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fvisibility=hidden --cuda-device-only -DMP_FRAME_ISA_SUBSET "$@" -S \
  -o /tmp/frame_q.s meltingpot_amd/csrc/frame.hip 2>&1 | grep -v "warning\|hip-link"
python3 - <<'PY'
import re
txt = open("/tmp/frame_q.s").read()
for m in re.finditer(r"\.name:\s+(_ZN\S*k_frame\S*)\n(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))
    short = re.sub(r"^_ZN\d+_GLOBAL__N_\d+k_frameI", "", name)[:50]
    print(f"{short:52s} sgpr spill {get('sgpr_spill_count'):4d}  vgpr {get('vgpr_count'):3d} spill {get('vgpr_spill_count'):3d}")
for m in re.finditer(r"^(_ZN\S*k_frame\S*):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
    body = m.group(2)
    short = re.sub(r"^_ZN\d+_GLOBAL__N_\d+k_frameI", "", m.group(1))[:50]
    n = len([l for l in body.splitlines() if re.match(r"\s+[sv]_|\s+(ds|global|buffer|flat)_", l)])
    print(f"{short:52s} {n:6d} instr")
m = re.search(r"^(_ZN\S*k_frameIN5stepk8NoTablesENS1_7NoSitesELi1E\S*):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M)
open("/tmp/frame_q_nt1.s", "w").write(m.group(2))
PY
