#!/bin/bash
R=rev_5e3a44d
profiles/ab.sh r04s12 "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" "$R 1921 1081 2048 noise8 slots 20" "base 1921 1081 2048 noise8 slots 20" "$R 1001 1001 4096 mixed slots 10" "base 1001 1001 4096 mixed slots 10" "$R 1366 768 4096 mixed slots 10" "base 1366 768 4096 mixed slots 10" "$R 1921 1081 6144 mixed slots 10" "base 1921 1081 6144 mixed slots 10" "$R 2999 2001 512 mixed slots 10" "base 2999 2001 512 mixed slots 10"
