#!/bin/bash
# Per-kernel register / LDS / spill report:  ./usage.sh file.hip
HERE=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$HERE/../../include -I$HERE \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | \
  grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size" | \
  sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | sed 's/  */ /g'
