"""Discriminator forward convs with and without the fused InstanceNorm column sums (cost of the statistics epilogue)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D
from microbench_disc import bench  # noqa
