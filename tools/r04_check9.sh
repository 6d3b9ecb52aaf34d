#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/prof_counters.sh E2 '[{}]' r04j/pmc_E2 2>&1 | grep -E "stack_" 
bash tools/prof_counters.sh E1 '[{}]' r04j/pmc_E1 2>&1 | grep -E "stack_"
