"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol
include/fami.h declares, and the host-side mirror of the reference's front door (registry, build_model,
module tree / state_dict keys, init statistics, error behaviour) behaves like the reference's
(posetimation/zoo/build.py:12-88, utils/utils_registry.py:14-76, Alignment_V15.py:27-111,185-248)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import fami_pose_amd as fp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from fami_pose_amd import _lib
    protos = _lib.parse_header()
    text = open(_lib.HEADER_PATH).read()
    declared = set(re.findall(r'\b(fami_\w+)\s*\(', re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)))
    declared.discard('fami_stream_t')
    assert declared == set(protos), declared ^ set(protos)      # the parser sees every prototype
    assert len(protos) >= 45
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), 'libfami_hip.so lacks %s' % name
    L = _lib.lib()
    assert L.cdll.fami_version().decode().startswith('fami-pose_amd')


def test_host_only_entry_points():
    """Entry points that do no device work are callable without a GPU: size queries and argument checks."""
    from fami_pose_amd._lib import lib, FamiError
    L = lib()
    # f32 fragment image [9][3][3][64][4] + (round 4) the pre-split image behind it: three bf16 planes = 1.5x the floats
    assert L.cdll.fami_packed_weight_elems(48, 48, 3, 3, 0) == 9 * 3 * 3 * 256 + 9 * 3 * 3 * 384
    assert L.cdll.fami_packed_weight_elems(48, 20, 3, 3, 0) == 9 * 2 * 3 * 256            # K % 16 != 0: no split image
    assert L.cdll.fami_packed_weight_elems(256, 64, 1, 1, 0) == 4 * 16 * 256 + 8 * 8 * 256      # 16- and 32-tile images
    assert L.cdll.fami_packed_weight_elems(17, 48, 1, 1, 0) == 3 * 2 * 256
    assert L.cdll.fami_conv2d_wgrad_workspace(20, 96, 72, 48, 48, 3, 3, 1, 1, 1) > 0
    assert L.cdll.fami_conv2d_wgrad_workspace(1, 8, 8, 8, 8, 3, 3, 3, 1, 1) == -1      # stride 3 unsupported
    assert L.cdll.fami_bn_workspace(48) > 0 and L.cdll.fami_shift_workspace(4) > 0
    with pytest.raises(FamiError, match='bad argument'):
        L.call('fami_conv2d_fwd_f32', None, None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 1, 1, 1, 0, 0, None)
    assert b'fami_conv2d_fwd_f32' in L.cdll.fami_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from fami_pose_amd import _lib
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.FamiError, match='no CPU fallback'):
        _lib._Lib()


# ------------------------------------------------------------------ registry / build_model
def test_registry_contract():
    from fami_pose_amd.zoo.registry import Registry
    r = Registry('X')

    @r.register()
    class A:
        pass

    class B:
        pass
    r.register(B)
    assert r.get('A') is A and r.get('B') is B
    with pytest.raises(AssertionError, match="already registered in 'X'"):
        r.register(B)
    with pytest.raises(KeyError, match="No object named 'C' found in 'X' registry"):
        r.get('C')
    for name in ('Alignment_V15', 'HRNet', 'HRNetPlus'):
        assert fp.MODEL_REGISTRY.get(name) is getattr(fp, name)


@pytest.fixture(scope='module')
def v15():
    torch.manual_seed(0)
    return fp.build_model(fp.default_cfg(48), fp.TRAIN_PHASE)


def test_build_model_phases_and_hyperparameters(v15):
    cfg = fp.default_cfg(48)
    assert v15.training and v15.is_train
    val = fp.build_model(fp.default_cfg(32, image_size=(192, 256)), fp.VAL_PHASE)
    assert (not val.training) and (not val.is_train)
    assert fp.get_model_hyperparameter(cfg) == 'bbox_1.25_rot_45_scale_0.65-1.35_MseLoss_1.0'
    cfg.MODEL.NAME = 'HRNet'
    assert fp.get_model_hyperparameter(cfg) == 'bbox_1.25_rot_45_scale_0.65-1.35'
    cfg.MODEL.NAME = 'nope'
    with pytest.raises(KeyError):
        fp.build_model(cfg, fp.TRAIN_PHASE)
    assert cfg.MODEL.EXTRA is cfg['MODEL']['EXTRA']           # attribute AND item access (hrnet.py:571 vs :590)


def test_state_dict_keys_equal_reference(v15):
    want = [ln.split() for ln in open(os.path.join(GOLD, 'g10_state_dict_keys.txt')) if ln.strip()]
    sd = v15.state_dict()
    assert [k for k, _, _ in want] == list(sd.keys())
    for k, shp, dt in want:
        assert ('x'.join(map(str, sd[k].shape)) or 'scalar') == shp and str(sd[k].dtype) == 'torch.' + dt, k
    n_h = sum(p.numel() for k, p in v15.named_parameters() if k.startswith('hrnet.'))
    n_all = sum(p.numel() for p in v15.parameters())
    assert (n_h, n_all - n_h) == (63595745, 1059459)


def test_init_weights_statistics_match_reference(v15):
    g = np.load(os.path.join(GOLD, 'g11_init_stats.npz'))
    sd = v15.state_dict()
    for key in g.files:
        name, stat = key.rsplit('.', 1)
        t = sd[name].double()
        if stat == 'std':
            ref = float(g[key])
            assert t.std().item() == pytest.approx(ref, rel=0.15), name     # same distribution, different draw
        elif stat == 'absmax':
            assert t.abs().max().item() == float(g[key]) == 0.0, name
        elif stat in ('min', 'max'):
            assert getattr(t, stat)().item() == float(g[key]) == 1.0, name
    assert abs(sd['hrnet.conv1.weight'].std().item() - 1e-3) < 2e-4       # N(0, 0.001^2) on every nn.Conv2d


def test_freeze_and_generalised_heads():
    m = fp.build_model(fp.default_cfg(48, freeze_backbone=True), fp.TRAIN_PHASE)
    assert not any(p.requires_grad for p in m.hrnet.parameters())
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 1059459
    m7 = fp.build_model(fp.default_cfg(48, image_size=(384, 512), num_sup=7), fp.TRAIN_PHASE)
    assert m7.sup_agg_block.layers[0].conv1.in_channels == 48 * 7
    assert m7.feat_global_offset_layers[7].in_features == 16 * 4 * 3
    m64 = fp.build_model(fp.default_cfg(64), fp.TRAIN_PHASE)
    assert m64.G == 16 and m64.dcn_offset_1.conv.out_channels == 18 * 16


def test_no_cpu_fallback(v15):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        v15(torch.zeros(1, 3, 384, 288), torch.zeros(1, 12, 384, 288))
    from fami_pose_amd.loss import JointMSELoss
    with pytest.raises(RuntimeError, match='HIP path only'):
        JointMSELoss()(torch.zeros(1, 17, 4, 4), torch.zeros(1, 17, 4, 4), torch.ones(1, 17, 1))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'fami-pose_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_product_init_matches_oracle_init():
    """bench.py initialises with fami_pose_amd.init.realistic_init_ (the product may not import the oracle); the parity
    tests use the oracle's copy.  Same seed -> same weights, on the same module tree."""
    from fami_pose_amd.init import realistic_init_
    from oracle import model as om
    a = fp.build_model(fp.default_cfg(32, image_size=(96, 128), num_sup=2), fp.TRAIN_PHASE)
    b = fp.build_model(fp.default_cfg(32, image_size=(96, 128), num_sup=2), fp.TRAIN_PHASE)
    realistic_init_(a, 11)
    om.realistic_init_(b, 11)
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_input_pipeline_host_logic_matches_reference_golden():
    """fami_pose_amd.data (crop transform, its inverse, joint transform, flip bookkeeping, visibility rule) against the
    vectors the reference's datasets/process modules produced (tests/golden/g13_input_pipeline.npz)."""
    from fami_pose_amd import data as D
    g = np.load(os.path.join(GOLD, 'g13_input_pipeline.npz'))
    for b in range(4):
        t = D.dark_get_affine_transform(g['centers'][b], g['scales'][b], float(g['rots'][b]), g['image_size'])
        assert np.array_equal(t, g['trans'][b])
        assert np.array_equal(D.dark_get_affine_transform(g['centers'][b], g['scales'][b], float(g['rots'][b]), g['image_size'], inv=1),
                              g['trans_inv'][b])
        mi = D.invert_affine(t)
        assert np.allclose(mi @ np.append(t @ np.array([7.0, 9.0, 1.0]), 1.0), [7.0, 9.0], atol=1e-8)
        fj, fv = D.fliplr_joints(g['joints'][b], g['vis'][b], 1280)
        assert np.array_equal(fj, g['flip_joints'][b]) and np.array_equal(fv, g['flip_vis'][b])
        j2, v2 = D.transform_joints(g['joints'][b], g['vis'][b], t, g['image_size'])
        for j in range(17):
            if g['vis'][b, j, 0] > 0:
                assert np.array_equal(j2[j, 0:2].astype(np.float32), g['pts'][b, j].astype(np.float32))
            x, y = j2[j, 0], j2[j, 1]
            outside = x < 0 or y < 0 or x > 288 or y > 384
            assert (v2[j] == 0).all() if outside else np.array_equal(v2[j], g['vis'][b, j])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        D.crop_clip(torch.zeros(2, 8, 8, 3, dtype=torch.uint8), [4, 4], [0.1, 0.1], 0, (8, 8))


def test_every_environment_switch_goes_through_the_one_parser():
    """The host layer reads FAMI_* switches only through fami_pose_amd/options.py, whose table is the complete list (name,
    default, meaning); kernel-routing state is not environment at all but the fami_route_t struct (include/fami_route.h)."""
    import glob
    import re
    from fami_pose_amd import options
    from fami_pose_amd._lib import Route
    pkg = os.path.dirname(os.path.abspath(options.__file__))
    used = set()
    for f in glob.glob(os.path.join(pkg, '**', '*.py'), recursive=True):
        src = open(f).read()
        if not f.endswith('options.py'):
            assert not re.search(r"os\.environ[^\n]*FAMI_", src), f          # no direct read anywhere else
        used |= set(re.findall(r"options\.(?:get|flag|number)\('(FAMI_[A-Z0-9_]+)'", src))
    assert used and used <= set(options.SWITCHES), sorted(used - set(options.SWITCHES))
    with pytest.raises(KeyError):
        options.get('FAMI_NOT_A_SWITCH')
    assert options.flag('FAMI_LANES') and not options.flag('FAMI_WGRAD_LANE')
    # the route struct mirrors the header: ints / longs only, first field its own size
    names = [n for n, _ in Route._fields_]
    assert names[0] == 'size' and len(names) > 60 and {'use_t6', 'dcn_bwd2', 'wg6_target', 'bn_small_elems'} <= set(names)
    hdr = open(os.path.join(os.path.dirname(pkg), 'include', 'fami_route.h')).read()
    assert all(re.search(r'\b%s;' % n, hdr) for n in names)
    # ... and the LIBRARY agrees with the header field by field: fami_route_init must produce the default each field's comment
    # documents (a stale object file compiled against an older layout passes the size check -- padding -- and shifts every field)
    from fami_pose_amd._lib import lib
    r = lib().new_route()
    doc = {m.group(1): int(m.group(2)) for m in re.finditer(r'\b(\w+);\s*/\* default (-?\d+)', hdr)}
    assert len(doc) >= len(names) - 1
    wrong = {n: (getattr(r, n), v) for n, v in doc.items() if getattr(r, n) != v}
    assert not wrong, wrong


def test_ablation_switches_are_refused_at_every_read(monkeypatch):
    """The WRONG-by-design ablation switches (options.WRONG) raise whenever they are READ without FAMI_ALLOW_WRONG=1 -- not only
    when the library loads: their values are read per Engine / Trainer, so one set later in a process must not slip through."""
    from fami_pose_amd import options
    for name in options.WRONG:
        monkeypatch.setenv(name, '1')
        monkeypatch.delenv('FAMI_ALLOW_WRONG', raising=False)
        with pytest.raises(RuntimeError):
            options.get(name)
        with pytest.raises(RuntimeError):
            options.flag(name)
        monkeypatch.setenv('FAMI_ALLOW_WRONG', '1')
        assert options.flag(name)
        monkeypatch.setenv(name, '0')
        monkeypatch.delenv('FAMI_ALLOW_WRONG', raising=False)
        assert not options.flag(name)
        monkeypatch.delenv(name)


def test_new_routes_take_the_environment_ab_switches(monkeypatch):
    """The A/B switches that are route fields (FAMI_T5, FAMI_WGS3_TARGET, ...) reach every route lib().new_route() hands out, not only
    the process default written at load: an Engine with a route of its own measures the kernels the environment names."""
    from fami_pose_amd._lib import lib
    L = lib()
    base = L.new_route()
    assert base.use_t5 == 1 and base.wgs3_target == 0
    monkeypatch.setenv('FAMI_T5', '0')
    monkeypatch.setenv('FAMI_WGS3_TARGET', '128')
    r = L.new_route()
    assert r.use_t5 == 0 and r.wgs3_target == 128
    monkeypatch.delenv('FAMI_T5')
    monkeypatch.delenv('FAMI_WGS3_TARGET')
    assert L.new_route().use_t5 == 1


def test_combined_backward_instances_cover_the_layer_shapes_of_the_path():
    """csrc/conv_pair.hip holds one combined (input gradient + weight gradient) kernel instance per layer shape: the plans are host
    code, so the table is checked here against every 3x3 stride-1 layer shape of BASELINE configs 2 / 3 / 5 (16-bit storage) and
    against the persistent split-product kernel's shapes (f32 storage) -- a plan change that moves a layer to another instance must
    show up as a failure here, not as a silent fall-back to two launches on the GPU box."""
    from fami_pose_amd._lib import lib
    L = lib().cdll
    L.fami_tune_reset()
    geo = lambda N, H, W, Ci, Co: (N, H, W, Ci, Co, 3, 3, 1, 1, 1)
    half = [(20, 96, 72, 48, 48), (20, 48, 36, 96, 96), (20, 24, 18, 192, 192), (20, 12, 9, 384, 384),      # W48, 5-frame clips, batch 4
            (24, 96, 72, 48, 48), (24, 48, 36, 96, 96), (24, 24, 18, 192, 192), (24, 12, 9, 384, 384),      # config 2: 3-frame, batch 8
            (4, 96, 72, 48, 48), (4, 96, 72, 96, 48), (4, 96, 72, 192, 48), (8, 96, 72, 48, 48),            # the head's aggregation blocks
            (20, 96, 72, 64, 64), (20, 48, 36, 128, 128), (20, 24, 18, 256, 256), (20, 12, 9, 512, 512)]    # HRNet-W64 (config 5)
    for s in half:
        assert L.fami_conv2d_bwd_pair_ok(*geo(*s)) == 1, s
    for s in [(20, 96, 72, 48, 48), (20, 48, 36, 96, 96), (24, 96, 72, 48, 48), (4, 96, 72, 96, 48), (8, 128, 96, 48, 48)]:
        assert L.fami_conv2d_bwd_pair_ok_f32(*geo(*s)) == 1, s
    # not combined by design: other geometries, the band kernel's f32 launches, maps no DMA-staged kernel takes
    assert L.fami_conv2d_bwd_pair_ok(20, 96, 72, 48, 48, 3, 3, 2, 1, 1) == 0 and L.fami_conv2d_bwd_pair_ok(20, 96, 72, 48, 48, 1, 1, 1, 0, 1) == 0
    assert L.fami_conv2d_bwd_pair_ok_f32(*geo(20, 24, 18, 192, 192)) == 0 and L.fami_conv2d_bwd_pair_ok(*geo(2, 32, 24, 48, 48)) == 0
    L.fami_conv_tune_lds(8998)
    try:
        assert L.fami_conv2d_bwd_pair_ok(*geo(20, 96, 72, 48, 48)) == 0
    finally:
        L.fami_tune_reset()
