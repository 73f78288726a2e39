#!/bin/bash
AB_ARGS="--n 768" AB_STEPS=2 bash profiles/tools/ab.sh default wave pre wavepre wavenopad nopad 2>&1 | grep "ms per\|zpass"
AB_ARGS="--n 200" AB_STEPS=5 bash profiles/tools/ab.sh default wave pre wavepre wavenopad nopad 2>&1 | grep "ms per\|zpass"
