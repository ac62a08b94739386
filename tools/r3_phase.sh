#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/phase_time.py && timeout -k 10 200 python3 tools/phase_time.py
