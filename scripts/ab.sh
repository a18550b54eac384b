#!/bin/bash
# usage: ab.sh "<libs>" "<bench args>" ...   ("" = the product library)
libs="$1"; shift
for rep in 1 2; do for lib in $libs; do if [ "$lib" = "cur" ]; then unset ILQR_AMD_LIB; else export ILQR_AMD_LIB=$PWD/ilqr_amd/lib/exp/$lib.so; fi
for a in "$@"; do timeout 300 python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['configs']; print('[$lib] $a', round(d['ms_per_step'],4), round(d['stages']['backward']['ms_per_launch'],4), 'late', round(c['late_solve']['late_ms_per_step'],4), round(c['late_solve']['stages_ms']['backward'],4))"; done; done; done
