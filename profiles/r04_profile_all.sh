#!/bin/bash
# rocprofv3 evidence of round 4 (kernel trace + separate PMC passes) for the bench workloads; see run_profile.sh
# usage: r04_profile_all.sh [tag] [part]   part 1: configs[1] noise8 + mixed; part 2: configs[2], configs[3]; part 3: DBDE16
T=${1:-r04}; P=${2:-1}
if [ "$P" == "1" ]; then bash profiles/run_profile.sh $T 2 noise8; bash profiles/run_profile.sh $T 2 mixed; fi
if [ "$P" == "2" ]; then bash profiles/run_profile.sh $T 3 mixed; bash profiles/run_profile.sh $T 4 mixed; fi
if [ "$P" == "3" ]; then bash profiles/run_profile_u16.sh $T; fi
