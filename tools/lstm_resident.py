#!/usr/bin/env python3
"""Print what a4nt_init's occupancy probe found: co-resident workgroups per persistent-sweep instantiation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a4nt_amd
from a4nt_amd import _lib

lib = _lib.load()
_lib.init_device(0)
for H in (64, 128, 256, 512, 1024):
    print('H=%4d' % H, {mt: lib.a4nt_lstm_resident_workgroups(0, mt, H) for mt in (1, 2, 4)},
          'geometry(B=128)', a4nt_amd.ops.lstm_geometry(128, H), 'geometry(B=256)', a4nt_amd.ops.lstm_geometry(256, H))
