#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/r3_tests2.sh
bash tools/r3_profile.sh
