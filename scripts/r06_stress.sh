#!/bin/bash
timeout 600 python scripts/stress_one_query.py 30000 2>&1 | tail -4
