#!/bin/bash
timeout 900 python tools/pf_repro2.py 400 2>&1 | grep -v amdgpu.ids
