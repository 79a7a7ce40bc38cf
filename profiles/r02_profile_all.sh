#!/bin/bash
T=${1:-r02}
bash profiles/run_profile.sh $T 2 noise8
bash profiles/run_profile.sh $T 2 mixed
bash profiles/run_profile.sh $T 3 mixed
bash profiles/run_profile.sh $T 4 mixed
