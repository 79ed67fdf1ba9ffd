#!/bin/bash
# tools/kernel_regs.sh <file.s> [symbol substring]: VGPRs / AGPRs / scratch / LDS / occupancy of every kernel of an assembly listing
# (hipcc -S --cuda-device-only) whose mangled name contains the substring.
awk -v pat="${2:-}" '/^_Z[A-Za-z0-9_]*:/ { name = $1 } /^; NumVgprs|^; NumAgprs|^; ScratchSize|^; LDSByteSize|^; Occupancy/ { if (index(name, pat)) { gsub(/^; /, ""); gsub(/ bytes\/workgroup.*/, ""); printf "%s  ", $0; if ($0 ~ /Occupancy/) printf "  %s\n", name } }' "$1"
