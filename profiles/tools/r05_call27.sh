#!/bin/bash
AB_ARGS="--n 768" AB_STEPS=2 bash profiles/tools/ab.sh default skipload 2>&1 | tail -15
AB_ARGS="--n 200" AB_STEPS=5 bash profiles/tools/ab.sh default percu16 skipload 2>&1 | tail -15
