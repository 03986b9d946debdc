#!/bin/bash
# Which combination makes the interpreter abort at exit ("double free or corruption") after an RCCL communicator was made?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out/dbg
run() { echo "== $1"; shift; timeout 120 python -c "$@" > /tmp/out.txt 2>&1; echo "rc=$?"; grep -i "double free\|corrupt\|Abort\|Error" /tmp/out.txt | head -3; }
run "comm create+close, no torch" "
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0); c.close()"
run "comm create, no close, no torch" "
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0)"
run "import torch first, comm create+close" "
import torch
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0); c.close()"
run "import torch + cuda init first, comm create+close" "
import torch; torch.cuda.init(); torch.zeros(1, device='cuda')
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0); c.close()"
run "library only (env create/close), no comm" "
import numpy as np
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
e = TradingEnvironment(num_trajectories=1024, seed=1); e.reset(); e.close()"
run "comm + allreduce on env stream, no torch" "
import numpy as np
from mbt_gym_amd.distributed import RcclCommunicator
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
e = TradingEnvironment(num_trajectories=1024, seed=1); e.reset()
c = RcclCommunicator(0, 1, 0)
print(e.allreduce_return_sums(c, [1.0, 2.0, 3.0]))
e.close(); c.close()"
run "comm, os._exit" "
import os
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0); c.close(); os._exit(0)"
run "env first, then torch imported, then comm" "
import numpy as np
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
e = TradingEnvironment(num_trajectories=1024, seed=1); e.reset()
import torch
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0)
print(e.allreduce_return_sums(c, [1.0, 2.0, 3.0]))
e.close(); c.close()"
run "comm first, then torch + torch tensor" "
from mbt_gym_amd.distributed import RcclCommunicator
c = RcclCommunicator(0, 1, 0)
import torch; print(torch.zeros(2, device='cuda').sum().item())
c.close()"
run "jit plugin env (hiprtc), no torch" "
import sys; sys.path.insert(0, 'tests')
import numpy as np
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
from mbt_gym_amd.rewards.RewardFunctions import DeviceExpressionReward
print('ok')"
