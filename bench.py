#!/usr/bin/env python
"""bench.py -- decoder hot-path throughput on MI355X (BASELINE.json metric: samples/sec, 6-cam T=8 900q
r50 704x256; plus the sampling kernel's HBM roofline fraction and a CPU native-PyTorch baseline).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the decoder hot path (6 shared-weight layers: scale-adaptive self attention,
adaptive spatio-temporal sampling, adaptive mixing, FFN, heads) over one batch of synthetic input whose FPN
features are already resident in HBM in the reference's own layout.  Ranks shard by sample (weak scaling: the
per-GPU batch is fixed); the only collective is the end-of-run metric all-reduce (RCCL), mirroring the
reference's end-of-eval result gather (val.py:132).  Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
CHILD_ARGS = []        # workload flags the profiled re-runs of this command (live_kernel_stats / live_pmc) have to repeat
PROFILE_EVERY = 5      # sampler launches are bracketed with HIP events in every 5th step of the timed region (see main)
sys.path.insert(0, ROOT)

from sparsebev_amd import runtime, synthetic as S                  # noqa: E402
from sparsebev_amd.parallel import SampleShard, init_distributed   # noqa: E402
from sparsebev_amd.transformer import SparseBEVTransformer         # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0      # best measured device copy rate in the same guide (SURVEY.md section 8d asks for both)
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA dense peak (same guide, matrix-core table)
MFMA_16BIT_PEAK_TFLOPS = 2500.0   # fp16 / bf16 dense MFMA peak (same table; the 2:1-sparsity figure is not used)
GEMM_WHAT = {
    'f16x3': 'fp32-class (default): fp32 operands as scaled fp16 hi + lo images, 3 products (hl, lh, hh), f32 accumulate on the 16-bit matrix core '
             '(csrc/gemm_bf16s.hip); max and rms error vs fp64 below the exact f32 MFMA kernels (tests/test_gpu_bf16s.py); mixing generator + out-projection only',
    'f16x4': 'the same with all 4 fp16 image products',
    'f32': 'exact f32-input MFMA kernels (gemm.hip / gemm_regtile.hip): the default of rounds 1-2',
    'bf16x6': 'fp32-class: hi + mid + lo bf16 images, 6 products, f32 accumulate (csrc/gemm_bf16s.hip)',
    'bf16x3s': '3 x bf16 split products (2^-16 class) on the gemm_bf16s.hip kernels',
    'bf16x3': '3 x bf16 split products, f32 accumulate (sbev_linear_bf16x3, round-2 kernels)'}
GEMM_PRODUCTS = {'f16x3': 3, 'f16x4': 4, 'bf16x6': 6, 'bf16x3s': 3, 'bf16x3': 3}
DEFAULT_GEMM = os.environ.get('SBEV_GEMM_MODE') or 'f16x3'
OUT8_MIN_ROWS = int(os.environ.get('SBEV_OUT8_MIN_ROWS', '1024')) or (1 << 30)      # rows from which the out-projection runs on 256-row tiles (csrc/gemm_bf16s.hip)


def pmc_profile(config, kernel='msmv_fwd_kernel'):
    """Counter record of `kernel` for this bench config from the committed rocprofv3 PMC summaries: separate --pmc
    FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum passes of this same command (tools/profile_pmc.sh;
    KiB units, FETCH_SIZE x2 gfx950 correction -- tools/pmc_summary.py).  PMC counters cannot be collected from inside
    the timed process, so this is the last committed profile of the SAME config: profiles/<round>_pmc_<config>.json
    (newest round wins).  Returns (record, file name) or (None, None)."""
    pdir = os.path.join(ROOT, 'profiles')
    best = (None, None)
    if os.path.isdir(pdir):
        legacy = {'c2': '_pmc_summary.json', 'c5': '_c5_pmc_summary.json'}.get(config)
        for fn in sorted(os.listdir(pdir)):
            if fn.endswith('_pmc_%s.json' % config) or (legacy and fn.endswith(legacy) and fn.split('_')[0] + legacy == fn):
                try:
                    k = json.load(open(os.path.join(pdir, fn)))['kernels'].get(kernel)
                    if k:
                        best = (k, fn)
                except Exception:      # noqa: BLE001
                    pass
    return best


def live_kernel_stats(config, gemm, timeout=120):
    """Average kernel durations of THIS run's decoder step by `rocprofv3 --kernel-trace --stats` (no counters: durations are not
    disturbed by counter collection), measured now: bench.py re-runs itself for 20 steps under the tracer and the stats CSV is
    parsed.  Returns {short kernel name: avg_us} or None (same guards as live_pmc)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('SBEV_BENCH_CHILD') == '1' or shutil.which('rocprofv3') is None:
        return None
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTX', 'ROCTRACER')) or k == 'HSA_TOOLS_LIB' for k in os.environ):
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_summary
    tmp = tempfile.mkdtemp(prefix='sbev_kt_', dir='/tmp')
    env = dict(os.environ, SBEV_BENCH_CHILD='1', TMPDIR='/tmp')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', tmp, '-o', 'b', '--',
               sys.executable, os.path.join(ROOT, 'bench.py'), '--config', config, '--gemm', gemm, '--steps', '20', '--warmup', '3',
               '--no-cpu-baseline', '--no-alt', '--no-detector', '--no-live-pmc'] + CHILD_ARGS
        r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout)
        csvs = [os.path.join(d, f) for d, _, fs in os.walk(tmp) for f in fs if f.endswith('kernel_stats.csv')]
        if r.returncode != 0 or not csvs:
            return None
        res = {}
        for row in csv.DictReader(open(csvs[0])):
            sh = pmc_summary.short(row['Name'])
            if sh is not None and sh not in res:
                res[sh] = round(float(row['AverageNs']) / 1e3, 2)
        return res or None
    except Exception:      # noqa: BLE001  (a measurement aid must never take the metric line down)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_pmc(config, timeout=90):
    """HBM-side byte counters of THIS run's kernels, measured now: bench.py re-runs itself for a few steps under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes -- the two do not fit one; --kernel-trace only beside
    them) on the same box and parses the counter CSVs exactly like tools/pmc_summary.py (KiB units, FETCH_SIZE x2 on gfx950).
    Returns {short kernel name: {'hbm_bytes_per_launch': ...}} or None when rocprofv3 is unavailable / a pass fails (the
    committed per-config profile is used then)."""
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('SBEV_BENCH_CHILD') == '1' or shutil.which('rocprofv3') is None:
        return None
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTX', 'ROCTRACER')) or k == 'HSA_TOOLS_LIB' for k in os.environ):
        return None          # this process is itself being profiled: no nested profiler (committed profile instead)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_summary
    tmp = tempfile.mkdtemp(prefix='sbev_pmc_', dir='/tmp')
    env = dict(os.environ, SBEV_BENCH_CHILD='1', TMPDIR='/tmp')
    res = {}
    try:
        per = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '-o', 'b', '--',
                   sys.executable, os.path.join(ROOT, 'bench.py'), '--config', config, '--steps', '3', '--warmup', '2',
                   '--no-cpu-baseline', '--no-alt', '--no-detector', '--no-live-pmc'] + CHILD_ARGS
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout)
            csvs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith('counter_collection.csv')]
            if r.returncode != 0 or not csvs:
                return None
            per[counter] = pmc_summary.per_kernel(csvs[0], counter)
        for name in set(per['FETCH_SIZE']) | set(per['WRITE_SIZE']):
            sh = pmc_summary.short(name)
            if sh is None:
                continue
            fk, n = per['FETCH_SIZE'].get(name, (0.0, 0))
            wk, _ = per['WRITE_SIZE'].get(name, (0.0, 0))
            if sh in res and res[sh]['launches_sampled'] >= n:
                continue
            res[sh] = {'launches_sampled': n, 'fetch_bytes_corrected_x2': int(2 * fk * 1024), 'write_bytes': int(wk * 1024),
                       'hbm_bytes_per_launch': int(2 * fk * 1024 + wk * 1024)}
        return res or None
    except Exception:      # noqa: BLE001  (a measurement aid must never take the metric line down)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


CONFIGS = {
    # name: (pyramid, Q, T, per-GPU batch, feature dtype)           -- SURVEY.md section 8 config table
    'c2': ('r50_704x256', 900, 8, 1, torch.float32),     # BASELINE.json configs[1]: the metric's config
    'c3': ('r50_704x256', 400, 8, 8, torch.float32),
    'c4': ('r101_1408x512', 900, 8, 4, torch.float32),
    'c1': ('r50_704x256', 100, 1, 1, torch.float32),
    # configs[4] at one sample per GPU: ViT-scale 1600x640 maps, 5 levels, bf16 feature STORAGE (fp32 math), channels-last
    # as a bf16 neck would emit them (2.1 GB of features per sample): the large-feature-map HBM stress
    'c5': ('eva02_1600x640', 900, 8, 1, torch.bfloat16),
    # the reference's own largest config (configs/vit_eva02_1600x640_trainval_future.py:54-58): 1600 queries, 15 frames (7 past +
    # 7 future), 8 sampling points, 5 levels; bf16 feature storage like c5 -- 3.9 GB of features per sample
    'c6': ('eva02_1600x640', 1600, 15, 1, torch.bfloat16, 8),
}


def cfg_fields(cfg):
    """(pyramid, Q, T, per-GPU batch, feature dtype, sampling points)"""
    return tuple(cfg[:5]) + ((cfg[5] if len(cfg) > 5 else 4),)


def build_model(T, L, device, P=4):
    torch.manual_seed(0)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=6, num_levels=L, num_classes=10,
                             code_size=10, pc_range=S.PC_RANGE)
    m.init_weights()
    S.randomize_zero_init(m, std=0.02, seed=0)      # query-dependent offsets / mixing weights / tau
    return m.to(device).eval()


def cpu_baseline(cfg, model_state, max_seconds=30.0, thread_sweep=(8, 16, 32, 64, 128, 256)):
    """The reference's native-PyTorch path (grid_sample sampler + eager ops), restated in oracle/ and
    validated against golden vectors, timed on this box's host cores.  Bounded sample: 1 warm-up + up to 3
    timed samples or ~max_seconds, whichever comes first."""
    from oracle import sparsebev_oracle as O       # checker / baseline only -- never on the product path
    pyr, Q, T, B, _, P = cfg_fields(cfg)
    ih, iw, sizes = S.PYRAMIDS[pyr]
    cores = os.cpu_count() or 1
    params = O.strip_prefix({k: v.detach().cpu().float() for k, v in model_state.items()})
    bbox, feat = S.make_queries(1, Q, seed=0)
    feats = S.make_features(1, T, sizes, seed=0)
    metas = S.make_img_metas(1, T, ih, iw)
    with torch.no_grad():
        # torch's intra-op pool does not scale to every core of a big host (256 threads is ~20x SLOWER than 32
        # here): calibrate the thread count on one decoder layer, keep the fastest -- the baseline at its best.
        best = None
        for nt in [n for n in thread_sweep if n <= cores] or [cores]:
            torch.set_num_threads(nt)
            O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1, num_points=P)
            t0 = time.perf_counter()
            O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=1, num_points=P)
            dt1 = time.perf_counter() - t0
            if best is None or dt1 < best[0]:
                best = (dt1, nt)
            if dt1 > 2.5 * best[0]:
                break
        threads = best[1]
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_points=P)
        warm = time.perf_counter() - t0
        n, t0 = 0, time.perf_counter()
        while n < 3 and (n == 0 or (time.perf_counter() - t0) + warm < max_seconds):
            O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_points=P)
            n += 1
        dt = time.perf_counter() - t0
    cores_used = threads
    return {'value': round(n / dt, 4), 'unit': 'samples/s', 'cores': cores_used, 'kind': 'port',
            'cpu_model': cpu_model(), 'host_cores': cores,
            'sample': '%d timed decoder samples (1 warm-up) at %s Q=%d T=%d bs=1, oracle grid_sample path, '
                      'torch intra-op threads=%d (fastest of a 1-layer sweep) on a %d-core host' % (n, pyr, Q, T, cores_used, cores)}


def gpu_quick(config, device, steps=30, warmup=5):
    """decoder samples/s of another config (same method as the timed region, fewer steps): used for the c1 leg next to its
    CPU baseline -- c1 is the reference's own CPU-runnable case (BASELINE.json configs[0]), never `value`."""
    pyr, Q, T, B, fdtype, P = cfg_fields(CONFIGS[config])
    ih, iw, sizes = S.PYRAMIDS[pyr]
    model = build_model(T, len(sizes), device, P)
    feats = S.make_features(B, T, sizes, seed=0, device=device, dtype=fdtype)
    bbox, qfeat = [t.to(device) for t in S.make_queries(B, Q, seed=0)]
    metas = S.make_img_metas(B, T, ih, iw)
    for _ in range(warmup):
        model(bbox, qfeat, list(feats), None, copy.deepcopy(metas))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(bbox, qfeat, list(feats), None, copy.deepcopy(metas))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return round(steps * B / dt, 2), model


def ops_sample_mix_supported(L, T, P=4):
    from sparsebev_amd import ops
    return os.environ.get('SBEV_NO_SAMPLE_MIX') != '1' and ops.sample_mix_supported(L, 64, P, T, 4)


def mfma_util():
    """MFMA-pipe utilisation per kernel from the committed PMC summary (tools/mfma_summary.py), {} when absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
    try:
        names = sorted(f for f in os.listdir(path) if f.endswith('_mfma_summary.json'))
        d = json.load(open(os.path.join(path, names[-1])))['kernels']
        return {k: v['mfma_util_pct'] / 100.0 for k, v in d.items()}
    except Exception:   # noqa: BLE001
        return {}


def detector_standin(args, T, L, Q, B, ih, iw, sizes, device, transformer):
    """Labelled secondary figure (never `value`): one streaming step of a whole detector = backbone + FPN stand-in on
    the 6 NEW images (fp16, channels-last), features copied into the per-frame ring, SparseBEVHead (query init, the
    decoder under test, output re-format) and the NMS-free decode.  Mirrors timing.py:77-96 of the reference."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
    import backbone_standin
    from sparsebev_amd.cache import FrameFeatureCache
    from sparsebev_amd.head import SparseBEVHead
    net = backbone_standin.build(device, num_outs=L)
    head = SparseBEVHead(num_classes=10, in_channels=256, num_query=Q, code_size=10,
                         transformer=dict(type='SparseBEVTransformer', embed_dims=256, num_frames=T, num_points=4, num_layers=6,
                                          num_levels=L, num_classes=10, code_size=10, pc_range=S.PC_RANGE),
                         bbox_coder=dict(type='NMSFreeCoder', post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                         pc_range=S.PC_RANGE, max_num=300, score_threshold=0.05, num_classes=10))
    head.transformer.load_state_dict(transformer.state_dict())
    with torch.no_grad():
        head.init_query_bbox.weight[:, 2] = 0.5
        # random-init class bias is -4.6 (sigmoid 0.01 < the 0.05 score threshold: the decode would compact NOTHING and the figure
        # would time an empty post-process); a trained head keeps a few hundred boxes -- bias 0 puts ~half the scores above it
        head.transformer.decoder.decoder_layer.cls_branch[-1].bias.zero_()
    head = head.to(device).eval()
    metas = S.make_img_metas(B, T, ih, iw)
    imgs = torch.rand(B * 6, 3, ih, iw, device=device) * 255
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
        dts = {f.dtype for f in net(imgs)}
    # the ring keeps the frames in the type the backbone emits (fp16 under autocast: taps are widened exactly inside the sampler --
    # DESIGN 10.9); a mixed-type neck falls back to the fp32 ring (frames widened on entry)
    ring = FrameFeatureCache(T, n_slots=T, dtype=dts.pop() if len(dts) == 1 and dts <= {torch.float16, torch.bfloat16} else torch.float32)

    def frame():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.float16):
            feats = net(imgs)
        ring.push([f.view(B, 6, *f.shape[1:]) for f in feats])

    for _ in range(T):
        frame()

    def step():
        frame()
        outs = head(ring.pyramid(), metas)
        return head.get_bboxes(outs, metas)

    for _ in range(max(3, args.warmup // 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = max(10, args.steps // 2)
    for _ in range(n):
        res = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    dtb = time.perf_counter() - t1
    return {'value': round(n * B / dt, 3), 'unit': 'samples/s', 'ms_per_step': round(1e3 * dt / n, 4),
            'backbone_ms': round(1e3 * dtb / n, 4), 'boxes_kept_last_step': int(sum(r[0].shape[0] for r in res)),
            'ring_dtype': str(ring.dtype).replace('torch.', ''),
            'what': 'LABELLED STAND-IN, not the metric: stock torch.nn ResNet-50 + FPN (random init, fp16 autocast, MIOpen) on the 6 new '
                    '%dx%d images per sample + frame ring (in the backbone\'s own storage type) + SparseBEVHead (this repo) + NMS-free decode, online mode, bs=%d; the reference '
                    'publishes 15.8 FPS for this pipeline on an RTX 3090 (README.md:28)' % (iw, ih, B)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--nhwc', action='store_true', help='features already channels-last in HBM (zero-copy input)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-live-pmc', action='store_true', help='do not re-run a few steps under rocprofv3 --pmc for roofline.traffic (use the committed per-config profile)')
    ap.add_argument('--online', action='store_true', help='streaming mode: per step only ONE new frame (6 images) is relayouted into the per-frame feature ring (cache.FrameFeatureCache); the other T-1 frames stay resident')
    ap.add_argument('--no-detector', action='store_true', help='skip the labelled detector-level stand-in figure (default N=1 c2 run reports it)')
    ap.add_argument('--detector', action='store_true', help='force the detector figure for other configs too: also report a LABELLED detector-level samples/s: stock-PyTorch ResNet-50 + FPN stand-in (tools/backbone_standin.py, fp16) on the 6 new images -> frame ring -> SparseBEVHead -> NMS-free decode (online mode, like the reference FPS)')
    ap.add_argument('--no-alt', action='store_true', help='skip the secondary bf16x3 measurement')
    ap.add_argument('--feat-dtype', default=None, choices=('fp32', 'fp16', 'bf16'),
                    help='feature STORAGE type (fp32 math throughout; default: the config\'s own -- fp32, bf16 for c5 / c6).  fp16 = what the reference\'s eval mode holds '
                         'before its out_fp32 cast (val.py:115): NCHW lists of it go through the 2-byte relayout inside the step unless --nhwc')
    ap.add_argument('--query-order', type=int, default=None, choices=(0, 1, 2),
                    help='fused gather + mixing items in the order of sbev_query_order (1) or in launch order (0); default: the library setting (SBEV_QUERY_ORDER). Results are bit-identical')
    ap.add_argument('--shuffle-queries', action='store_true', help='permute the query rows (a trained head does not keep the BEV raster order of its initialisation): robustness A/B for --query-order')
    ap.add_argument('--overlap', type=int, default=0, help='0 = single stream (default); 1 = generator GEMM + classification branch on a second stream; 2 = classification branch only')
    ap.add_argument('--dense-relayout', action='store_true', help='A/B: relayout the WHOLE NCHW pyramid every step (rounds 1-5) instead of only the feature units the '
                                                                   'sample points read (on-demand relayout, csrc/layout.hip; bit-identical outputs)')
    ap.add_argument('--gemm', default=DEFAULT_GEMM, choices=sorted(runtime.GEMM_MODES),
                    help='the two big mixing GEMMs: f16x3 = fp32-class scaled fp16 hi + lo split, 3 products (default); f32 = exact f32-input MFMA; '
                         'bf16x6 = hi + mid + lo bf16 images, 6 products; f16x4 = 4 fp16 products; bf16x3s / bf16x3 = 3 bf16 products (2^-16 class)')
    args = ap.parse_args()

    for flag, val in (('--feat-dtype', args.feat_dtype), ('--query-order', args.query_order)):
        if val is not None:
            CHILD_ARGS.extend([flag, str(val)])
    if args.dense_relayout or os.environ.get('SBEV_NO_SPARSE_RELAYOUT'):
        args.dense_relayout = True
        runtime.lazy_relayout(False)
    for flag, on in (('--nhwc', args.nhwc), ('--online', args.online), ('--shuffle-queries', args.shuffle_queries), ('--dense-relayout', args.dense_relayout)):
        if on:
            CHILD_ARGS.append(flag)
    torch.set_grad_enabled(False)     # inference benchmark, like the reference's timing.py / val.py (with grad enabled the
                                      # module takes its differentiable path, as the reference's nn.Module would)
    rank, world, device = init_distributed(args.gpus)
    if world > 1:
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))      # N host processes share the box: no 256-thread pools each
    cfg = CONFIGS[args.config]
    pyr, Q, T, B, fdtype, P_cfg = cfg_fields(cfg)
    cfg_nhwc = fdtype != torch.float32          # c5 / c6: bf16 maps channels-last, as a bf16 neck would emit them
    if args.feat_dtype is not None:
        fdtype = {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}[args.feat_dtype]
    ih, iw, sizes = S.PYRAMIDS[pyr]
    L = len(sizes)

    model = build_model(T, L, device, P_cfg)
    model.decoder.gemm_mode = args.gemm
    model.decoder.overlap = args.overlap
    shard = SampleShard(rank, world)
    # per-rank synthetic inputs (seed = rank), generated on the device and left resident
    feats = S.make_features(B, T, sizes, seed=rank, device=device, dtype=fdtype)
    if args.nhwc or cfg_nhwc:
        args.nhwc = True
        feats = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
    bbox, qfeat = S.make_queries(B, Q, seed=rank)
    if args.shuffle_queries:
        perm = torch.randperm(Q, generator=torch.Generator().manual_seed(1234))
        bbox, qfeat = bbox[:, perm].contiguous(), qfeat[:, perm].contiguous()
    bbox, qfeat = bbox.to(device), qfeat.to(device)
    metas = S.make_img_metas(B, T, ih, iw)
    if args.query_order is not None:
        runtime.query_order(args.query_order)

    if args.online:
        from sparsebev_amd.cache import FrameFeatureCache
        ring = FrameFeatureCache(T, n_slots=T, dtype=fdtype)      # (fp16 / bf16 storage: the frames stay in their type)
        per_frame = [[f[:, t * 6:(t + 1) * 6].contiguous() for f in feats] for t in range(T)]
        for fr in reversed(per_frame):
            ring.push(fr)
        tick = [0]

        def step():
            ring.push(per_frame[tick[0] % T])         # the new frame's 6 images (NCHW, as the neck emits them)
            tick[0] += 1
            return model(bbox, qfeat, ring.pyramid(), None, metas)
    else:
        def step():
            return model(bbox, qfeat, list(feats), None, metas)

    for _ in range(args.warmup):
        step()
    # The timed region runs un-instrumented: with the same input tensors every step the module replays ONE captured hipGraph per
    # step (feature relayout + 6 layers; runtime.StepGraphs), and HIP events cannot be read back from a graph.  The kernel
    # timings for the roofline fields come from extra EAGER steps right after it (same inputs, same cache state).
    fused_cfg = ops_sample_mix_supported(L, T, P_cfg) and not (L == 5 and fdtype == torch.float32 and os.environ.get('SBEV_NO_FUSE_L5F32'))     # the runtime's own rule (csrc/decoder.hip)
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    checksum = 0.0
    for _ in range(args.steps):
        cls, box = step()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = time.perf_counter() - t0
    runtime.check_pair_faults()       # a timed step with a lost pair hand-off would be an invalid (and ~1 s slow) step: raise, never report it
    # host time to enqueue ONE step while the queue has room (the upload ring is 8 deep: issuing more than that many steps ahead
    # simply waits for the GPU, which is what the round-2 figure measured)
    t1 = time.perf_counter()
    for _ in range(6):
        step()
    host_issue = (time.perf_counter() - t1) / 6
    torch.cuda.synchronize()
    rt = model.decoder._runtime
    graph_info = {'replays': rt.step_graphs.replays, 'captures': rt.step_graphs.captures} if rt is not None else None
    launches_per_layer = rt.launches_per_layer(B, Q) if rt is not None else None
    # HIP events around the gather launches, on their stream, in every 5th of the following eager steps: the two records
    # around a launch leave ~5.6 us of idle stream each.  The decoder step runs the gather FUSED with the adaptive-mixing
    # kernel where the fused launch covers the shape (kind 3); otherwise the stand-alone sampler (kind 0) is what it launches.
    runtime.profile_stride(PROFILE_EVERY)
    runtime.profile_sampler(8 if fused_cfg else 1)
    for _ in range(min(30, max(10, args.steps))):
        step()
    torch.cuda.synchronize()
    fused_ms = sorted(runtime.read_kernel_ms(3)) if fused_cfg else []
    kernel_ms = [] if fused_cfg else sorted(runtime.read_sampler_ms())
    # a few extra steps outside the timed region: the two mixing GEMMs bracketed, and -- when the timed steps ran the fused
    # launch -- the same decoder step with the sampler as its own launch (fusion off), so that the stand-alone sampling
    # kernel of BASELINE.json's metric is timed in the decoder's cache state on the same inputs
    runtime.profile_stride(1)
    if fused_cfg:                                # (own phase: event records around the neighbouring GEMM launches would inflate it)
        runtime.fuse_sample_mix(False)
        for _ in range(3):                       # the unfused step touches a workspace region the fused one never did: warm it
            step()
        runtime.profile_stride(PROFILE_EVERY)
        runtime.profile_sampler(1)
        for _ in range(min(20, 2 * args.steps)):
            step()
        torch.cuda.synchronize()
        kernel_ms = sorted(runtime.read_sampler_ms())
        runtime.fuse_sample_mix(True)
        runtime.profile_stride(1)
    runtime.profile_sampler(6)
    for _ in range(min(10, args.steps)):
        step()
    torch.cuda.synchronize()
    gemm_ms = [sorted(runtime.read_kernel_ms(k)) for k in (1, 2)]
    runtime.profile_sampler(False)
    checksum = float(cls.double().abs().sum().item() + box.double().abs().sum().item())

    # secondary measurements (rank 0, N = 1 only): the same steps in the other modes of the two big mixing GEMMs -- first of all the
    # exact f32-input MFMA kernels -- reported next to, never instead of, `value` (which is the mode of --gemm, by default the library's)
    alt = None
    if world == 1 and args.gemm == DEFAULT_GEMM and not args.no_alt:
        alt = {}
        for mode in [m for m in ('f32', 'f16x3', 'bf16x6', 'bf16x3s', 'bf16x3') if m != args.gemm]:
            what = GEMM_WHAT[mode]
            model.decoder.gemm_mode = mode
            try:
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    cls3, box3 = step()
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t1
                runtime.profile_sampler(6)
                for _ in range(min(5, args.steps)):
                    step()
                torch.cuda.synchronize()
                g_ms = [runtime.read_kernel_ms(k) for k in (1, 2)]
                runtime.profile_sampler(False)
                alt[mode] = {'gemm': what, 'value': round(args.steps * B / dt3, 3), 'unit': 'samples/s', 'ms_per_step': round(1e3 * dt3 / args.steps, 4),
                             'generator_us': round(1e3 * sum(g_ms[0]) / len(g_ms[0]), 2) if g_ms[0] else None,      # (None: this mode's kernel is not bracketed)
                             'out_proj_us': round(1e3 * sum(g_ms[1]) / len(g_ms[1]), 2) if g_ms[1] else None,
                             'max_abs_dev_vs_default_layer0': round(float(max((cls3[0] - cls[0]).abs().max(), (box3[0] - box[0]).abs().max())), 8)}
            except Exception as e:      # noqa: BLE001  (a secondary figure must never take the metric line down)
                alt[mode] = {'error': repr(e)[:300]}
            finally:
                model.decoder.gemm_mode = args.gemm

    # ... and the same step EAGER (no graph replay) on the exact kernels: the figure that stays comparable with rounds 1-2 and with an
    # eager reference loop
    eager_f32 = None
    if world == 1 and args.gemm == DEFAULT_GEMM and not args.no_alt:
        sg, model.decoder.static_graph, model.decoder.gemm_mode = model.decoder.static_graph, False, 'f32'
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            eager_f32 = {'value': round(args.steps * B / dt, 3), 'unit': 'samples/s', 'ms_per_step': round(1e3 * dt / args.steps, 4),
                         'what': 'the same steps enqueued eagerly (no hipGraph replay), exact f32-input MFMA kernels for the two mixing GEMMs'}
        except Exception as e:      # noqa: BLE001
            eager_f32 = {'error': repr(e)[:300]}
        finally:
            model.decoder.static_graph, model.decoder.gemm_mode = sg, args.gemm

    detector = None
    if world == 1 and (args.detector or (args.config == 'c2' and not args.no_detector and not args.online)):
        try:
            detector = detector_standin(args, T, L, Q, B, ih, iw, sizes, device, model)
        except Exception as e:      # noqa: BLE001  (a labelled secondary figure must never take the metric line down)
            detector = {'error': repr(e)[:300]}

    # the one collective: metric all-reduce (MAX of elapsed, SUM of samples / checksum) over RCCL
    elapsed_max, samples, checksum_sum, elapsed_min = shard.reduce_metrics(elapsed, args.steps * B, checksum, per_rank=True)

    if rank == 0:
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        smp = model.decoder.decoder_layer.sampling
        G_, P_, Cg_ = smp.num_groups, smp.num_points, 256 // smp.num_groups
        npts = B * T * G_ * Q * P_                                 # B' * Q * P sampled points per launch
        sf = 4 if fdtype == torch.float32 else 2
        bytes_per_pt = L * 4 * Cg_ * sf + 12 + 4 * L + Cg_ * 4     # SURVEY.md section 8d byte model
        alg_bytes = npts * bytes_per_pt
        alg_gbps = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # physical HBM-side rate: PMC bytes per launch (fabric-side L2 requests = HBM + Infinity-Cache traffic) of the last
        # committed profile of THIS config over the live HIP-event time -- a fraction of the 8 TB/s peak by construction.
        # The section-8d algorithmic rate prices every tap as a miss and therefore exceeds the pin rate whenever taps hit in
        # L2: it is reported beside it with the implied hit fraction, never as `frac`.
        # (the committed profiles are of the config's own storage type: with --feat-dtype only counters measured now apply)
        own_dtype = args.feat_dtype is None or {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}[args.feat_dtype] == cfg_fields(cfg)[4]
        pmc, pmc_file = pmc_profile(args.config) if own_dtype else (None, None)
        live = live_pmc(args.config) if (world == 1 and not args.no_live_pmc) else None
        live_src = 'measured in this run: bench.py re-ran 5 steps of this config under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes; x2 gfx950 correction), bytes per launch'
        if live and 'msmv_fwd_kernel' in live:
            pmc = dict(pmc or {}, **live['msmv_fwd_kernel'])          # keeps the committed profile's L2 hit ratio beside the live bytes
            pmc_file = None
        traffic = pmc['hbm_bytes_per_launch'] if pmc else None
        hbm_gbps = traffic / (avg_ms * 1e-3) / 1e9 if (traffic and avg_ms > 0) else None
        out = {
            'metric': 'decoder samples/sec (6-cam T=%d %dq %s, 6 layers, features resident in HBM)' % (T, Q, pyr),
            'value': round(samples / elapsed_max, 3), 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed_max / args.steps, 4),
            'host_issue_ms_per_step': round(1e3 * host_issue, 4),
            # the two comparison figures up front (short fields: they survive a truncated record): the graph-replayed step on the exact
            # f32-input MFMA kernels, and that step enqueued eagerly
            'exact_f32': ({'value': alt['f32']['value'], 'ms_per_step': alt['f32']['ms_per_step']} if (alt and 'value' in alt.get('f32', {})) else None),
            'eager_f32': ({'value': eager_f32['value'], 'ms_per_step': eager_f32['ms_per_step']} if (eager_f32 and 'value' in eager_f32) else None),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32' if fdtype == torch.float32 else ('bf16' if fdtype == torch.bfloat16 else 'fp16') + '-storage/f32-math') + ('' if args.gemm == 'f32' else ' (the two mixing GEMMs: %s = %s)' % (args.gemm, GEMM_WHAT[args.gemm])),
            'data': 'synthetic',
            'config': {'workload': '%s: %s, %d queries, T=%d, bs=%d per GPU, 6 decoder layers, random-init weights, '
                                   '%s feature input%s' % (args.config, pyr, Q, T, B, 'online ring: 1 new NCHW frame relayouted per step, T-1 cached' if args.online else ('NHWC zero-copy' if args.nhwc else 'NCHW (reference layout, relayout inside the step%s)' % (': dense' if args.dense_relayout else ': on demand -- only the feature units the sample points read')),
                                                           '' if own_dtype else ', %s feature storage' % args.feat_dtype),
                       'global_batch': B * world, 'parallelism': 'sample-sharded x%d' % world,
                       'launches_per_layer': launches_per_layer, 'step_graph': graph_info,
                       'query_order': bool(runtime._STATE['order']), 'shuffled_queries': bool(args.shuffle_queries),
                       'relayout': None if (args.nhwc or args.online) else ('dense' if args.dense_relayout else 'on-demand'),
                       'checksum': checksum_sum},
            # per-rank spread (weak scaling: every rank runs the same per-GPU batch): slowest / fastest rank's own rate
            'per_rank_samples_per_s': {'min': round(args.steps * B / elapsed_max, 3), 'max': round(args.steps * B / elapsed_min, 3)},
            'roofline': {'kernel': 'msmv_fwd_kernel (adaptive sampling gather)', 'bound': 'hbm',
                         'achieved': round(hbm_gbps, 1) if hbm_gbps else None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': round(hbm_gbps / HBM_PEAK_GBPS, 4) if hbm_gbps else None,
                         'frac_of_measured_copy_peak': round(hbm_gbps / HBM_COPY_GBPS, 4) if hbm_gbps else None,
                         'traffic': traffic,
                         'traffic_source': (live_src if pmc_file is None else 'profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, bytes per launch, same config'
                                            % pmc_file) if pmc else 'no PMC profile for this config',
                         'achieved_is': 'PMC bytes per launch / live HIP-event time per launch (physical, <= peak); achieved_algorithmic is the '
                                        'SURVEY 8d byte model (every tap priced as a miss) over the same time',
                         'achieved_algorithmic': round(alg_gbps, 1),
                         'algorithmic_over_peak': round(alg_gbps / HBM_PEAK_GBPS, 4),
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'cache_served_fraction': round(1.0 - traffic / alg_bytes, 4) if traffic else None,
                         'l2_hit_ratio_pmc': pmc.get('l2_hit_ratio') if pmc else None,
                         'launches': len(kernel_ms), 'avg_us': round(avg_ms * 1e3, 2),
                         'event_sampling': ('HIP events around the stand-alone sampler launches of every %dth of %d extra decoder steps run with the fusion off right after '
                                            'the timed region (inside it the gather runs fused with the mixing kernel: roofline_fused)' % (PROFILE_EVERY, min(20, 2 * args.steps)))
                                           if fused_ms else 'HIP events around the sampler launches of every %dth of the eager steps run right after the timed region (which replays a captured graph)' % PROFILE_EVERY},
        }
        # the tracer's view of the same kernels (no event records, no launch gaps): this command re-run now under rocprofv3 --kernel-trace --stats
        kt = live_kernel_stats(args.config, args.gemm) if (world == 1 and not args.no_live_pmc) else None
        if kt and 'msmv_fwd_kernel' in kt:
            out['roofline']['avg_us_rocprof'] = kt['msmv_fwd_kernel']
            if traffic:
                out['roofline']['frac_rocprof'] = round(traffic / (kt['msmv_fwd_kernel'] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        if fused_ms:
            # the launch the timed steps really run: gather + adaptive mixing in one kernel.  Algorithmic bytes = the sampler's
            # gather reads (no [B,Q,G,T*P,C] write any more) + the item's dynamic parameters + its mixed output
            f_avg = sum(fused_ms) / len(fused_ms)
            items = B * Q * G_
            f_alg = npts * (L * 4 * Cg_ * sf + 12 + 4 * L) + items * ((Cg_ * Cg_ + 128 * T * P_) * 4 + 128 * Cg_ * 4)
            fp, fp_file = pmc_profile(args.config, 'adaptive_mixing_kernel') if own_dtype else (None, None)
            if live and 'adaptive_mixing_kernel' in live:
                fp, fp_file = live['adaptive_mixing_kernel'], None
            f_traffic = fp['hbm_bytes_per_launch'] if fp else None
            out['roofline_fused'] = {'kernel': 'adaptive_mixing_kernel<RT, true, L, FT> (gather + adaptive mixing, one launch)', 'bound': 'hbm',
                                     'achieved': round(f_traffic / (f_avg * 1e-3) / 1e9, 1) if f_traffic else None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                     'frac': round(f_traffic / (f_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if f_traffic else None,
                                     'traffic': f_traffic, 'traffic_source': (live_src if fp_file is None else 'profiles/%s' % fp_file) if fp else 'no PMC profile for this config',
                                     'achieved_algorithmic': round(f_alg / (f_avg * 1e-3) / 1e9, 1), 'algorithmic_bytes_per_launch': f_alg,
                                     'launches': len(fused_ms), 'avg_us': round(f_avg * 1e3, 2),
                                     'event_sampling': 'HIP events around the fused launches of every %dth of the eager steps run right after the timed region (which replays a captured graph)' % PROFILE_EVERY}
            if kt and 'adaptive_mixing_kernel' in kt:
                out['roofline_fused']['avg_us_rocprof'] = kt['adaptive_mixing_kernel']
            # the kernel the timed steps spend the most time in (the stand-alone sampler above is BASELINE.json's metric kernel; inside
            # the step its gather runs fused): first-class, same numbers as roofline_fused
            rf = out['roofline_fused']
            out['roofline']['dominant_kernel'] = {'kernel': 'adaptive_mixing_kernel (gather + adaptive mixing, one launch)', 'bound': 'hbm',
                                                  'avg_us': rf['avg_us'], 'achieved': rf['achieved'], 'frac': rf['frac'], 'traffic': rf['traffic'],
                                                  'share_of_step': round(6 * rf['avg_us'] * 1e-3 / (1e3 * elapsed_max / args.steps), 3)}
        # the kernels that dominate the step by TIME are the two mixing GEMMs (MFMA-bound, exact fp32): same live HIP-event
        # measurement, priced against the f32-input MFMA peak; PMC MFMA-pipe utilisation from profiles/ when present
        if args.gemm != 'f32' and all(gemm_ms):
            # the split-operand kernels: arithmetic done = nprod image products on the 16-bit matrix core; `fp32_equiv` prices the GEMM itself
            D_, Pin_, Pout_ = 256, T * P_, 128
            npr = GEMM_PRODUCTS[args.gemm]
            fl_g = 2.0 * B * Q * D_ * (G_ * (Cg_ * Cg_ + Pin_ * Pout_))
            fl_o = 2.0 * B * Q * D_ * (G_ * Pout_ * Cg_)
            g_us, o_us = (sum(ms) / len(ms) * 1e3 for ms in gemm_ms)
            out['roofline_mfma'] = [
                {'kernel': name, 'bound': 'mfma', 'achieved': round(npr * fl / (us * 1e-6) / 1e12, 1), 'peak': MFMA_16BIT_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                 'frac': round(npr * fl / (us * 1e-6) / 1e12 / MFMA_16BIT_PEAK_TFLOPS, 4), 'image_products': npr,
                 'fp32_equiv_tflops': round(fl / (us * 1e-6) / 1e12, 1), 'launches': len(ms), 'avg_us': round(us, 2), 'algorithmic_flop_per_launch': fl}
                for name, fl, us, ms in (('%s (mixing parameter generator, %s)' % ('gemm_f16s_gen_ws_kernel' if GEMM_PRODUCTS[args.gemm] in (3, 4) and args.gemm != 'bf16x3' else 'gemm_bf16s_gen3_kernel', args.gemm), fl_g, g_us, gemm_ms[0]),
                                         ('gemm_bf16s_out%s_kernel (mixing out-projection, split-K, %s)' % (('8' if B * Q >= OUT8_MIN_ROWS else '4') if args.gemm.startswith('f16') else '3', args.gemm), fl_o, o_us, gemm_ms[1]))]
            # the fp32-class emulation may be the default once generator + out-projection <= 2 x 60 us at config 2
            out['gemm_gate'] = {'generator_us': round(g_us, 2), 'out_proj_us': round(o_us, 2), 'sum_us': round(g_us + o_us, 2),
                                'measured': 'HIP events around the two launches of 10 eager steps right after the timed region (an upper bound: the records '
                                            'include the launch gap)', 'config': args.config}
            gk = (kt or {}).get('gemm_f16s_gen_ws_kernel', (kt or {}).get('gemm_bf16s_gen3_kernel'))
            if gk:
                ok = kt.get('gemm_bf16s_out8_kernel', kt.get('gemm_bf16s_out4_kernel', kt.get('gemm_bf16s_out3_kernel')))
                out['gemm_gate'].update({'generator_us_rocprof': gk, 'out_proj_us_rocprof': ok,
                                         'sum_us_rocprof': round(gk + ok, 2) if ok else None,
                                         'rocprof': 'rocprofv3 --kernel-trace --stats of 23 steps of this command re-run now on this box: AverageNs of the two kernels'})
        if args.gemm == 'f32' and all(gemm_ms):
            D_, Pin_, Pout_ = 256, T * P_, 128
            flops = 2.0 * B * Q * D_ * (G_ * (Cg_ * Cg_ + Pin_ * Pout_))          # generator; the out-projection has G*Pout*Cg*D = the same at Pin = 32
            flops2 = 2.0 * B * Q * D_ * (G_ * Pout_ * Cg_)
            util = mfma_util()
            out['roofline_mfma'] = [
                {'kernel': name, 'bound': 'mfma', 'achieved': round(fl / (sum(ms) / len(ms) * 1e-3) / 1e12, 1), 'peak': MFMA_F32_PEAK_TFLOPS,
                 'unit': 'TFLOP/s', 'frac': round(fl / (sum(ms) / len(ms) * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                 'launches': len(ms), 'avg_us': round(sum(ms) / len(ms) * 1e3, 2), 'algorithmic_flop_per_launch': fl,
                 'mfma_pipe_util_pmc': util.get(key)}
                for name, key, fl, ms in (('gemm_nt_f32_strip_kernel (mixing parameter generator)', 'gemm_nt_f32_strip_kernel<false>', flops, gemm_ms[0]),
                                          ('gemm_nt_f32_regtile_kernel (mixing out-projection, split-K)', 'gemm_nt_f32_regtile_kernel', flops2, gemm_ms[1]))]
        if alt is not None:
            out['alt_gemm'] = alt
        if eager_f32 is not None:
            out['eager_f32_detail'] = eager_f32
        if detector is not None:
            out['detector_standin'] = detector
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, model.state_dict())
            out['gpu_over_cpu'] = round(out['value'] / out['cpu_baseline']['value'], 1)
            if args.config == 'c2':
                # the CPU baseline at c1 too (BASELINE.json configs[0], the reference's own CPU-runnable case), with
                # the HIP path at the same shape beside it
                try:
                    c1_gpu, c1_model = gpu_quick('c1', device)
                    c1 = cpu_baseline(CONFIGS['c1'], c1_model.state_dict(), max_seconds=10.0, thread_sweep=(4, 8, 16, 32))
                    c1['gpu_value'] = c1_gpu
                    c1['gpu_over_cpu'] = round(c1_gpu / c1['value'], 1)
                    out['cpu_baseline']['c1'] = c1
                except Exception as e:      # noqa: BLE001
                    out['cpu_baseline']['c1'] = {'error': repr(e)[:300]}
        print(json.dumps(out))
    shard.shutdown()


if __name__ == '__main__':
    main()
