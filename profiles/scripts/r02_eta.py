"""Sweep of the inexact-Newton forcing term of the polish (experiment; env overrides)."""
import os, subprocess, sys
for emax, coef, pw in ((0.1, 1, 0.5), (0.03, 1, 0.5), (0.01, 1, 0.5), (0.01, 0.1, 0.5), (0.003, 1, 0.5), (0.01, 1, 1.0), (0.001, 1, 0.5)):
    env = dict(os.environ, SCORE_NEWTON_ETA_MAX=str(emax), SCORE_NEWTON_ETA_COEF=str(coef), SCORE_NEWTON_ETA_POW=str(pw))
    print(f"--- eta = min({emax}, {coef} |g|^{pw})", flush=True)
    subprocess.run([sys.executable, "profiles/scripts/r02_quick.py"], env=env)
