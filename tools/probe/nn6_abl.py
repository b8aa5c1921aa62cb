import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
for a in ("", "1", "2", "4", "8", "16", "32", "63"):
    env = dict(os.environ, NN6_ONE="1")
    if a:
        env["RECMAN_HIP_LIB"] = os.path.join(root, "build", "exp", f"librecman_a{a}.so")
    out = subprocess.run([sys.executable, os.path.join(here, "nn6_time.py")], env=env, capture_output=True, text=True)
    print(f"ABL={a or 0}:", out.stdout.strip().splitlines()[-1][:75] if out.stdout.strip() else out.stderr[-300:], flush=True)
