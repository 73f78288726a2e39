#!/bin/bash
AB_ARGS="--n 768" AB_STEPS=1 bash profiles/tools/ab.sh nopad sk1 sk2 sk4 sk8 sk15 2>&1 | grep -v "^xpass\|^ypass\|collapse"
