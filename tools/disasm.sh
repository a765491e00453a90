#!/bin/bash
# gfx950 ISA of one build: tools/disasm.sh <extra -D flags...>  -> /tmp/rt_disasm/<tag>.s (tag = RT_DISASM_TAG or "head")
cd "$(dirname "$0")/.." || exit 1
tag=${RT_DISASM_TAG:-head}
mkdir -p /tmp/rt_disasm
flags=$(python3 -c "import __graft_entry__ as g; print(' '.join(f for f in g.HIPFLAGS if f not in ('-shared', '-fPIC')))")
/opt/rocm/bin/hipcc $flags "$@" --cuda-device-only -S ray_tracer_amd/csrc/rt_device.hip -o /tmp/rt_disasm/$tag.s
