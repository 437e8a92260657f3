#!/bin/bash
# A/B of the config-5 training step on ONE box: "NAME=VALUE ..." environment strings, each timed twice, interleaved.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "$@"; do
    echo -n "[$cfg] "
    env $cfg timeout 300 python scripts/time_train_step.py 2>&1 | grep "train step"
  done
done
