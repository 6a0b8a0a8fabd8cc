#!/bin/bash
set -o pipefail
bash tests/gpu_r03d.sh || exit 1
bash tests/gpu_r03e.sh
