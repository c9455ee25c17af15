#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/exp/r3_tests2.sh
bash tools/exp/r3_profile.sh
