"""pmc_traffic_raw.txt (tools/profile_round.sh) -> the JSON record bench.py reads for roofline.traffic (profiles/rNN_pmc_traffic.json).
Each kernel entry is stamped with the git blob hash of its source file: bench.py drops the number when the source has changed."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _git_blob_sha  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = "centernet-pytorch-lightning_amd/csrc/"
# rocprofv3 kernel name pattern -> (bench.py kernel name, source file, note)
KERNELS = {
    "conv3x3s1_kernel<unsigned short, 128, 64, 8, 1>": ("conv3x3s1_kernel<bf16,128,64,8>", CSRC + "conv3x3.hip",
                                                      "halo-tile 3x3 kernel, launch mix of the >= 128-input-channel layers and the head data gradients"),
    "conv3x3_ws_kernel<64, false, 0, true, 0>": ("conv3x3_ws_kernel<64>", CSRC + "conv3x3_ws.hip",
                                              "weight-stationary kernel, ReLU variant = the three 64->256 head convs: 134 MB in (x 1.27 halo, x 4 channel blocks through L2) + 537 MB out"),
    "dcn_bwd_dom_kernel<64>": ("dcn_bwd_dom_kernel<64>", CSRC + "dcn_fused.hip", "offset/mask gradient of the 64-output-channel DCN layers"),
    "dcn_dom_bm_kernel<64, true>": ("dcn_dom_bm_kernel<64>", CSRC + "dcn_dom_bm.hip", "offset/mask gradient of the DCN layers with 64 output channels (gathered corner pixels + dot2; round 5)"),
    "dcn_dx_bm_kernel<2>": ("dcn_dx_bm_kernel<2>", CSRC + "dcn_bm.hip", "data gradient of the 64->64 DCN layers"),
    "dcn_fwd_bm_kernel<2>": ("dcn_fwd_bm_kernel<2>", CSRC + "dcn_bm.hip", "forward of the 64->64 DCN layers"),
    "dcn_fwd_b2_kernel<false>": ("dcn_fwd_b2_kernel", CSRC + "dcn_b2.hip", "forward of the 64->64 DCN layers, 16x16-tile kernel (round 6): 134 MB x (x 1.9 halo through L2) + 134 MB offsets + 134 MB y algorithmic"),
    "dcn_fwd_b2_kernel<true>": ("dcn_fwd_b2_kernel<MB>", CSRC + "dcn_b2.hip", "forward of the 128->64 / 256->64 DCN layers (launch mix)"),
    "dcn_wgrad_bm_kernel": ("dcn_wgrad_bm_kernel", CSRC + "dcn_bm.hip", "weight gradient of the DCN layers (launch mix)"),
    "topk_map128_kernel": ("topk_map128_kernel<true>", CSRC + "topk_stream.h", "B=64, C=80, 128x128 fp32 maps: 335.5 MB algorithmic read (SURVEY 8d)"),
    "bn_bwd_apply_kernel<unsigned short": ("bn_bwd_apply_kernel<bf16>", CSRC + "bn.hip", "launch mix of all BN layers"),
    "conv3x3_c16r_kernel<1, 1, 2>": ("conv3x3_c16r_kernel<1,1,AFF>", CSRC + "conv_c16.hip",
                                     "level0 forward 16->16 @512^2 with the stem's BN + ReLU applied on load: 537 MB in + 537 MB out algorithmic"),
    "conv3x3_c16r_kernel<1, 1, 0>": ("conv3x3_c16r_kernel<1,1,0>", CSRC + "conv_c16.hip",
                                     "level0 data gradient 16->16 @512^2 with the BN-backward statistics hook: 537 MB dy + 537 MB x in, 537 MB out algorithmic"),
    "conv3x3_c16r_kernel<2, 2, 2>": ("conv3x3_c16r_kernel<2,2,AFF>", CSRC + "conv_c16.hip",
                                     "level1 forward 16->32 stride 2: 537 MB in + 268 MB out algorithmic"),
    "stem7_rows_kernel": ("stem7_rows_kernel", CSRC + "wgrad_c16.hip", "stem 3->16 @512^2: 201 MB fp32 image in + 537 MB out algorithmic"),
    "pack_weight_batch_kernel": ("pack_weight_batch_kernel", CSRC + "conv_igemm.hip", "170 MB fp32 source (every weight read once per packing mode) + 86 MB bf16 out algorithmic"),
}
raw = [l.strip() for l in open(sys.argv[1]) if l.strip()]
vals = {}
for l in raw:
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE) (.+) launches (\d+) avg_kib ([0-9.e+-]+)", l)
    if m:
        vals[(m.group(2), m.group(1))] = (int(m.group(3)), float(m.group(4)))
out = {"recipe": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --no-cpu-baseline --no-probe --no-inference --no-extras --steps 3 --warmup 1 ; "
                 "same with --pmc WRITE_SIZE (separate passes, counters only). Both counters are in KiB. gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md: "
                 "FETCH_SIZE reports half of wide streaming reads -> doubled. WRITE_SIZE calibrated in round 1 on the 64->256 head conv launch: 524288 KiB reported = its 512 MiB output exactly.",
       "raw": raw, "kernels": {}}
for pat, (name, src, note) in KERNELS.items():
    f, w = vals.get((pat, "FETCH_SIZE")), vals.get((pat, "WRITE_SIZE"))
    if not f or not w or not f[0]:
        continue
    fb, wb = int(f[1] * 1024 * 2), int(w[1] * 1024)
    out["kernels"][name] = {"source": src, "source_blob_sha": _git_blob_sha(os.path.join(ROOT, src)), "launches_sampled": f[0],
                            "fetch_size_kib_raw_avg": round(f[1], 1), "fetch_bytes_per_launch": fb, "write_size_kib_avg": round(w[1], 1),
                            "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb, "note": f"rocprofv3 name {pat}; {note}"}
print(json.dumps(out, indent=1))
