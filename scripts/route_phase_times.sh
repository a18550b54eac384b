# experiment helper: phase times of the persistent routes (LIB = an alternative build of the library)
for F in ${ROUTES:-1 4}; do for FL in ${FLAGS:-0}; do
echo "== FUSED=$F flags=$FL"
ILQR_AMD_LIB=$LIB ILQR_AMD_FUSED=$F timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --flags $FL 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:(v['kernel'],round(v['ms_per_launch'],4)) for k,v in d['stages'].items()})"
done; done
