"""Summarise a rocprofv3 --kernel-trace results.db: a per-FAMILY roll-up (every kernel template is assigned to exactly one family, the
roll-up covers 100 % of the kernel time) followed by the per-kernel table of EVERY template (no cut-off: VERDICT r5 #8 -- the top-40
tables of rounds 1-5 left four templates of the dominant family out).
usage: python tools/prof_summary.py <results.db> [steps]      steps: training steps in the trace (default: launches of k_bce*)

Families follow bench.py's `families_ms_per_step` names; reduction kernels that several families share (k_reduce16, k_wgrad_reduce*)
are listed as their own line instead of being guessed into one of them."""
import re
import sqlite3
import sys

FAMILIES = [      # first match wins
    ("k_conv3_bx3 (fwd+dgrad)  [k_conv3_bx3 | k_conv3_ws | k_conv3_sp | k_conv3_spd]", r"k_conv3_bx3<|k_conv3_ws<|k_conv3_sp<|k_conv3_spd<"),
    ("k_wgrad3_bx3 (+ box-sum GEMM: k_spw_*, column scatter)", r"k_wgrad3_bx3<|k_spw_|k_scatter_cols"),
    ("k_conv3_thin_h (fwd+dgrad)  [+ k_conv3_thin_sp / k_conv3_thin_spd: decoder.blocks.4.conv1 in its sub-pixel forms]", r"k_conv3_thin_h<|k_conv3_thin_sp"),
    ("k_wgrad_thin_h", r"k_wgrad_thin_h<"),
    ("fp32-MFMA 3x3 (k_conv_mfma<3>, k_conv_mfma16, k_wgrad_mfma<3>, k_wgrad_mfma16)", r"k_conv_mfma<3|k_conv_mfma16|k_wgrad_mfma<3|k_wgrad_mfma16"),
    ("pointwise fwd+dgrad (k_conv_mfma<1>, k_pw_stream, k_pw3, k_conv1_ksplit, k_pwb_*)", r"k_conv_mfma<1|k_pw_stream<|k_pw3<|k_conv1_ksplit|k_ksplit|k_pwb_"),
    ("pointwise wgrad (k_wgrad_mfma<1>, k_pw3_wgrad)", r"k_wgrad_mfma<1|k_pw3_wgrad"),
    ("shared weight-gradient reductions (k_reduce16, k_wgrad_reduce*)", r"k_reduce16|k_wgrad_reduce"),
    ("k_irt_* (fused expand+dw)", r"k_irt_"),
    ("k_dw_* (depthwise)", r"k_dw_|k_dwconv"),
    ("k_stem_*", r"k_stem_"),
    ("k_head_*", r"k_head_"),
    ("BatchNorm finalize (forward)", r"k_bn_finalize|k_bn_tail|k_bn_rows_prereduce"),
    ("BatchNorm backward (reduce / small / finalize)", r"k_bn_bwd"),
    ("filter packing (k_pack*)", r"k_pack"),
    ("loss / Adam / casts / adds / other elementwise", r"k_bce|k_adam|k_cast|k_add|k_downsum|k_fill|k_sum_f32|k_threshold|k_thrconf|k_pred|k_gather|k_maxpool|k_upsample|k_clip|k_band_ratio|k_trimmed|k_rs_|k_percentile|k_order_stat"),
    ("mag1c", r"k_mag1c|k_lc_|k_li_|k_valid_mask|k_scatter"),
    ("torch / runtime kernels", r"at::|elementwise_kernel|vectorized|Memset|memcpy|__amd_rocclr"),
]

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else sum(r[1] for r in rows if "k_bce" in r[0])
steps = max(steps, 1)
print(f"# rocprofv3 --kernel-trace summary of {db}")
print(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; {steps} training steps in the trace "
      f"=> {tot/1e6/steps:.3f} ms of kernels and {sum(r[1] for r in rows)/steps:.0f} dispatches per step")
fam = {}
for n, cnt, s, a, mn, mx in rows:
    key = next((f for f, pat in FAMILIES if re.search(pat, n)), "unassigned")
    d = fam.setdefault(key, [0, 0.0, 0])
    d[0] += cnt; d[1] += s; d[2] += 1
print(f"# ---- per family (all {len(rows)} kernel templates, 100 % of the time)")
print(f"# {'family':100s} {'templates':>9s} {'calls/step':>10s} {'ms/step':>9s} {'pct':>6s}")
for k, (cnt, s, nt) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"# {k:100s} {nt:9d} {cnt/steps:10.1f} {s/1e6/steps:9.3f} {100*s/tot:6.2f}")
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for n, cnt, s, a, mn, mx in rows:
    short = n[:88]
    print(f"{short:90s} {cnt:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
