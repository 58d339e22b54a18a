#!/usr/bin/env python3
"""Emits the "what ships" table of DESIGN.md from a kernel-stats CSV of the one-stream train step (tools/profile_round.sh) and of the
16-bit inference group (tools/pmc_infer.sh): kernel -> source file:line -> role -> launches and time on that trace.
    python tools/what_ships.py profiles/<tag>_kernel_stats_serial.csv profiles/<tag>_infer_group_bf16_kernel_stats.csv [forwards]"""
import csv
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "wind-downscaling-gan_amd" / "csrc"
ROLE = [
    (r"wdg_igemm_kernel<128, 128", "implicit-GEMM conv fwd / dgrad, 128x128 tile (G: 8x8s2 23->128, 4x4s2, 3x3 128->512 gates; D: wide layers)"),
    (r"wdg_igemm_kernel<128, 64", "implicit GEMM 128x64 (D 7x7s3 32->64 fwd + LN; G 3x3 128->64)"),
    (r"wdg_dgrad_s3_kernel", "D 7x7s3 32->64 data gradient + LayerNorm backward below it: dy patch in LDS, nine residue classes per workgroup"),
    (r"wdg_dgrad_s3_(pack|finish)_kernel", "its weights in fragment order / its parameter sums"),
    (r"wdg_igemm_kernel<256, 32", "implicit GEMM 256x32 (1x1 column GEMMs of the upsample block; thin data gradients)"),
    (r"wdg_igemm_kernel<256, 16", "implicit GEMM 256x16 (thin data gradients)"),
    (r"wdg_igemm_kernel<64, 64", "implicit GEMM 64x64, LDS-DMA loop (D's small maps 29^2 .. 3^2)"),
    (r"wdg_igemm_kernel<128, 80", "implicit GEMM 128x80 (G 16->160 layer)"),
    (r"wdg_igemm_kernel<128, 160", "implicit GEMM 128x160"),
    (r"wdg_igemm_kernel<64, 128", "implicit GEMM 64x128 (+ LN backward)"),
    (r"wdg_wgrad_kernel", "weight gradient (rows = tap x channel group, pixel-split + fixed-order reduce)"),
    (r"wdg_wgrad_thin_kernel", "weight gradient of thin 3x3 full-resolution layers (+ bias row)"),
    (r"wdg_wgrad_reduce", "second stage of the split weight gradients"),
    (r"wdg_igemm_reduce", "split-K second stages (+ LN forward / backward)"),
    (r"wdg_conv_halo1_kernel", "persistent thin 3x3 conv, 16 channels in (G 16->2 output conv, D 16->16 + LN)"),
    (r"wdg_conv_halo_kernel", "halo-tile conv (<= 64 output channels on large maps)"),
    (r"wdg_convlstm1_fwd_mfma", "D's 5->16 single-timestep ConvLSTM forward (MFMA gates, cell on accumulators)"),
    (r"wdg_convlstm1_fwd_kernel", "D's 2->2 single-timestep ConvLSTM forward"),
    (r"wdg_convlstm1_bwd_kernel", "fused single-timestep ConvLSTM backward (gate recompute, dx, fused weight gradient)"),
    (r"wdg_convlstm1_wgrad_reduce", "sum of the fused ConvLSTM weight-gradient partials"),
    (r"wdg_convln_", "D's 2->16 conv + LeakyReLU + LayerNorm fused forward / recomputing backward"),
    (r"wdg_upconv_gather", "column-form upsample + 5x5 transposed conv: gather pass (fp32)"),
    (r"wdg_upconv_col_kernel", "column-form backward of the upsample block"),
    (r"wdg_bn_", "BatchNorm statistics finalize / apply / backward"),
    (r"wdg_ln_", "LayerNorm passes that are not fused"),
    (r"wdg_prep_", "batched spectral-norm power iteration + weight repack"),
    (r"wdg_philox", "Philox4x32-10 normal / uniform draws"),
    (r"wdg_input_assemble", "generator input [image | noise | 0] in one pass (fp32 or operand format)"),
    (r"wdg_adam_tf", "TF-form Adam on the flat buffers"),
    (r"wdg_lstm_fwd|wdg_lstm_bwd", "ConvLSTM cell pointwise (unfused routes)"),
    (r"wdg_copy_pixels|wdg_copy_channels|wdg_permute", "layout copies at the API edge"),
    (r"wdg_conv_patch_h16_kernel", "16-bit conv with the input patch in LDS (G's convs, gate conv, recurrent step with cell epilogue)"),
    (r"wdg_upconv_fused_h16", "16-bit upsample + 5x5 transposed conv, column form in one launch"),
    (r"wdg_conv_thin16_h16", "16-bit 16->2 output conv"),
    (r"wdg_dense_gap", "Dense(1) + GlobalAveragePooling head (+ LN backward)"),
    (r"wdg_lerp|wdg_sumsq|wdg_colsum|wdg_segment", "gradient-penalty interpolate / norms / column sums"),
]


def locate(name):
    base = re.match(r"[\w:]*?(wdg_\w+)", name.replace("(anonymous namespace)::", ""))
    if not base:
        return ""
    r = subprocess.run(["grep", "-n", "-E", rf"\b{base.group(1)}\(", "-r", str(CSRC), "--include=*.hip", "--include=*.h"], capture_output=True, text=True)
    for ln in r.stdout.splitlines():
        if "__global__" in ln or "__launch_bounds__" in ln:
            f, no = ln.split(":")[:2]
            return f"`csrc/{Path(f).name}:{no}`"
    first = r.stdout.splitlines()[:1]
    return f"`csrc/{Path(first[0].split(':')[0]).name}:{first[0].split(':')[1]}`" if first else ""


def table(path, per, unit):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"| kernel | source | role | launches per {unit} | mean us | ms per {unit} | share |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        ms = float(r["TotalDurationNs"]) / per / 1e6
        if ms < 0.02:
            continue
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
        role = next((t for pat, t in ROLE if re.search(pat, name)), "")
        out.append(f"| `{name[:70]}` | {locate(name)} | {role} | {int(r['Calls']) / per:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {ms:.3f} | {100 * float(r['TotalDurationNs']) / tot:.1f} % |")
    out.append(f"| **total** | | | | | **{tot / per / 1e6:.2f}** | |")
    return "\n".join(out)


if __name__ == "__main__":
    print("### Train step (one-stream schedule, rocprofv3 kernel trace, 2 steps)\n")
    print(table(sys.argv[1], 2, "step"))
    if len(sys.argv) > 2:
        fw = int(sys.argv[3]) if len(sys.argv) > 3 else 9
        print("\n### 16-bit inference forward (rocprofv3 kernel trace, eager launches)\n")
        print(table(sys.argv[2], fw, "forward"))
