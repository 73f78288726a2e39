#!/bin/bash
timeout 600 python3 profiles/tools/r05_slab_data_check.py 2>&1 | tail -6
