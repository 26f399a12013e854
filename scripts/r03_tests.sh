#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --durations=3 "$@" 2>&1 | tail -40
