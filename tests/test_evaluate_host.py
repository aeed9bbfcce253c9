"""Eval driver host logic against the reference's own PoseTrack JSON writer: the fixture
tests/golden/g14_posetrack_json.json holds synthetic predictions / boxes / annotation stubs and the files that
datasets/zoo/posetrack/PoseTrack_Alignment.py:883-1017 (`evaluate`, imported by oracle/gen_golden.py::g14_posetrack_json)
wrote for them -- PoseTrack17 (8-digit frames, 1-based) and PoseTrack18 (6-digit, 0-based) naming, frames without a
detection, two people in one frame, both annotation-file styles."""
import json
import os

import numpy as np

from fami_pose_amd import evaluate as ev

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g14_posetrack_json.json')


def test_json_writer_matches_reference(tmp_path):
    cases = json.load(open(GOLD))['cases']
    assert len(cases) == 2
    for k, c in enumerate(cases):
        annot = tmp_path / ('annot%d' % k)
        annot.mkdir()
        for fname, data in c['annotations'].items():
            (annot / fname).write_text(json.dumps(data))
        out = tmp_path / ('out%d' % k)
        written = ev.write_posetrack_results(np.array(c['preds']), np.array(c['boxes']), c['filenames_map'], str(annot),
                                             str(out), is_posetrack18=c['is_posetrack18'], phase='validate')
        got = {os.path.basename(p): json.load(open(p)) for p in written}
        assert sorted(got) == sorted(c['written'])
        for fn, want in c['written'].items():
            assert got[fn] == want, fn                      # identical structure AND float values (same float64 arithmetic)
        assert all(os.path.dirname(p).endswith('val_set_json_results') for p in written)
        # spot checks of what the reference produces: 15 PoseTrack joints, empty frames carry the dummy detection
        first = next(iter(c['written'].values()))['annolist']
        n_pts = [len(r['annopoints'][0]['point']) for el in first for r in el['annorect']]
        assert set(n_pts) <= {1, 15} and 15 in n_pts and 1 in n_pts


def test_accumulator_layout():
    """core fn :283-309: filenames_map counts images in arrival order (a name seen twice keeps both rows), boxes are
    (center, scale, prod(scale*200), score)."""
    acc = ev.EvalAccumulator(5, 17)
    rng = np.random.RandomState(0)
    for names in (['a/b/v1/00000001.jpg', 'a/b/v1/00000002.jpg', 'a/b/v1/00000001.jpg'], ['a/b/v2/00000001.jpg', 'a/b/v2/00000003.jpg']):
        n = len(names)
        c, s = rng.rand(n, 2).astype(np.float32), rng.rand(n, 2).astype(np.float32) + 0.5
        acc.add(rng.rand(n, 17, 2), rng.rand(n, 17, 1), rng.rand(n, 17, 2), rng.rand(n, 17, 1), c, s, rng.rand(n), names)
        last = (c, s)
    assert acc.idx == 5 and acc.filenames_map['a/b/v1/00000001.jpg'] == [0, 2] and acc.filenames_map['a/b/v2/00000003.jpg'] == [4]
    assert np.allclose(acc.all_boxes[3:, 4], np.prod(last[1] * 200, 1)) and np.allclose(acc.all_boxes[3:, 0:2], last[0])
    acc.add_accuracy(0, 0.5, 10)
    acc.add_accuracy(0, 1.0, 30)
    assert acc.accuracy(0) == 0.875 and acc.accuracy(1) == 0.0
