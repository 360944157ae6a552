#!/bin/bash
# usage: tools/match_bench.sh <n> <lib> <nb> : prints Gpairs/s, ms and frac of int8 peak of the match leg
SIFT_HIP_LIB=$2 SIFT_MATCH_NB=$3 timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --match-n $1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['match']
print('$2 nb=$3', d['value'], d['ms'], d['roofline']['frac'])"
