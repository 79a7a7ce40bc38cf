#!/bin/bash
# rocprofv3 evidence of round 3 (kernel trace + separate PMC passes) for the bench workloads; see run_profile.sh
T=${1:-r03}
bash profiles/run_profile.sh $T 2 noise8
bash profiles/run_profile.sh $T 2 mixed
bash profiles/run_profile.sh $T 3 mixed
bash profiles/run_profile.sh $T 4 mixed
bash profiles/run_profile_u16.sh $T
