#!/bin/bash
# Newton polish: chain factors recomputed when more than N cones changed activity since the last factorisation
# (SCORE_REFACTOR_FLIPS; default 0 = on any change, -1 = every iteration): solve time, Newton and PCG iterations
# on the headline problem and on a 4-robot trial
for n in -1 0 5 20 100; do
  echo "== refactor above $n flips"
  SCORE_REFACTOR_FLIPS=$n python3 profiles/scripts/r03_newton_queue.py 2>&1 | grep "^[0-9]"
  SCORE_REFACTOR_FLIPS=$n python3 profiles/scripts/r03_newton_queue.py 4 2>&1 | grep "^[0-9]"
done
