#!/bin/bash
# Registers, spills and occupancy of every trace_kernel instantiation, from the compiler's remarks.
# Usage: tools/kernel_resources.sh [pattern]     (runs `make asm` in ray-tracer_amd/csrc)
cd "$(dirname "$0")/../ray-tracer_amd/csrc" || exit 1
make asm 2>&1 | awk '
/Function Name:/ { if (name != "") print name, out; name = ""; out = ""
                   if (match($0, /trace_kernelILb[01]ELi[0-9]ELi[0-9]/)) { s = substr($0, RSTART + 14, RLENGTH - 14); gsub(/ELi/, " ", s); split(s, p, " ")
                       name = "split=" substr(p[1], 1, 1) " filter=" p[2] " variant=" p[3] } }
name != "" && /VGPRs:/ && !/Spill/ { out = out " vgpr=" $(NF-1) }
name != "" && /TotalSGPRs:/ { out = out " sgpr=" $(NF-1) }
name != "" && /Occupancy/ { out = out " occ=" $(NF-1) }
name != "" && /SGPRs Spill:/ { out = out " sgpr_spill=" $(NF-1) }
name != "" && /VGPRs Spill:/ { out = out " vgpr_spill=" $(NF-1) }
name != "" && /LDS Size/ { out = out " lds=" $(NF-1) }
END { if (name != "") print name, out }' | grep -E "${1:-.}"
