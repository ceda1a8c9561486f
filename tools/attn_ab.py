"""Attention kernel time (hipEvent pair around the launch, hg_profile_*) on the ViT-B/16 shape; with an experiments build
(HG_LIB_PATH=ab/exp.so) HG_ATTN_MODE knocks parts out: 1 no key loop, 2 no K/V staging, 4 no stores.
    NSEQ=256 L=197 HEADS=12 ROUNDS=6 python tools/attn_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
h = _lib.lib().hg_create(0)
n_seq, L, heads = int(os.environ.get("NSEQ", 256)), int(os.environ.get("L", 197)), int(os.environ.get("HEADS", 12))
qkv = torch.randn(n_seq * L, 3 * heads * 64, device="cuda")
out = torch.empty(n_seq * L, heads * 64, device="cuda")
ts = []
for it in range(int(os.environ.get("ROUNDS", 6)) + 2):
    def call():
        rc = _lib.lib().hg_test_attention(h, qkv.data_ptr(), None, None, n_seq, L, heads, int(os.environ.get("CAUSAL", 0)),
                                          out.data_ptr(), None)
        assert rc == 0, rc
    _, recs = _lib.profile(h, _lib.HG_PROF_ALL, 4, call)
    if it >= 2:
        ts.append(recs[0][4] * 1e3)
print(f"attention n_seq={n_seq} L={L} heads={heads}: median {statistics.median(ts):.1f} us (min {min(ts):.1f})")
