"""Print per-stage kernel times for a few configurations (GPU box helper)."""
import sys, os, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for args in sys.argv[1:]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--extras-out", "/tmp/stage_times_extras.json"] + args.split(), capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        x = json.load(open("/tmp/stage_times_extras.json"))  # (the per-stage table lives in the extras file since round 6)
        print(args, "| value %.3e ms/step %.3f |" % (d["value"], d["ms_per_step"]), {k: round(v["ms_per_launch"], 4) for k, v in x["stages"].items()})
    except Exception as e:
        print(args, "FAILED", e, out.stdout[-500:], out.stderr[-1500:])
