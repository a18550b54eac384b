# experiment helper: phase times of the persistent routes of the headline workload (LIB = an alternative build of the library)
#   ROUTES="1 2 3" FLAGS="0 16" LIB=path/to/lib.so bash scripts/route_phase_times.sh
for F in ${ROUTES:-1}; do for FL in ${FLAGS:-0}; do
echo "== route=$F flags=$FL"
ILQR_AMD_LIB=$LIB timeout 300 python bench.py --no-cpu-baseline --extras-out /tmp/rpt_extras.json --flags $FL --route $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=json.load(open('/tmp/rpt_extras.json')); print(d['ms_per_step'], {k:(v['kernel'],round(v['ms_per_launch'],4)) for k,v in x['stages'].items()})"
done; done
