"""pipeline.run_stream's scheduling -- lanes, the host-half threads, ordering, bounded look-ahead, error paths -- with the GPU call
replaced by a stand-in (no GPU needed; the real thing runs in tests/test_gpu_pipeline.py)."""
import time
import weakref

import pytest

from focalsv_amd import pipeline


class _Result:
    lines = []


class _Pending:
    def __init__(self, alive):
        self.res = _Result()
        alive.add(self.res)

    def finish(self):
        time.sleep(0.001)
        return self.res


@pytest.fixture
def fake(monkeypatch):
    state = {"alive": weakref.WeakSet(), "lanes": set()}

    def launch(ctx, batch, **kw):
        state["lanes"].add(ctx)
        time.sleep(0.003)
        return _Pending(state["alive"])

    monkeypatch.setattr(pipeline, "launch_hot_path", launch)
    return state


def test_order_lanes_and_bounded_lookahead(fake):
    seen, most = [], [0]

    def on_result(i, r):
        seen.append(i)
        most[0] = max(most[0], len(fake["alive"]))

    out = pipeline.run_stream(list("abcde"), [object()] * 120, on_result=on_result, keep_results=False)
    assert out == [] and seen == list(range(120))
    assert fake["lanes"] == set("abcde")
    assert most[0] <= 4 * 5 + 5 + 3          # the look-ahead window + the lanes + the few in the finisher's / caller's hands


def test_results_kept_iterators_static_and_empty(fake):
    assert len(pipeline.run_stream(list("ab"), [object()] * 9)) == 9
    assert len(pipeline.run_stream(list("ab"), iter([object()] * 7))) == 7
    assert len(pipeline.run_stream(list("abc"), [object()] * 8, static=True, stagger=0.001)) == 8
    assert pipeline.run_stream(list("ab"), []) == []


def test_failures_come_back_to_the_caller(monkeypatch):
    n = [0]

    def launch(ctx, batch, **kw):
        n[0] += 1
        if n[0] == 7:
            raise RuntimeError("launch failed")
        return _Pending(weakref.WeakSet())

    monkeypatch.setattr(pipeline, "launch_hot_path", launch)
    with pytest.raises(RuntimeError):
        pipeline.run_stream(list("abc"), [object()] * 40)

    class Bad:
        def finish(self):
            raise ValueError("finish failed")

    monkeypatch.setattr(pipeline, "launch_hot_path", lambda ctx, batch, **kw: Bad())
    with pytest.raises(ValueError):
        pipeline.run_stream(list("abc"), [object()] * 40)

    def producer():
        yield object()
        raise KeyError("producer failed")

    monkeypatch.setattr(pipeline, "launch_hot_path", lambda ctx, batch, **kw: _Pending(weakref.WeakSet()))
    with pytest.raises(KeyError):
        pipeline.run_stream(list("ab"), producer())


def test_on_result_failure_stops_the_stream_and_leaves_no_thread(fake):
    """an exception out of on_result (a failing gather, KeyboardInterrupt) used to leave the lanes spinning in take()"""
    import threading

    def on_result(i, r):
        if i == 2:
            raise RuntimeError("gather failed")

    t0 = time.time()
    with pytest.raises(RuntimeError, match="gather failed"):
        pipeline.run_stream(list("ab"), iter([object()] * 1000), on_result=on_result, keep_results=False)
    assert time.time() - t0 < 20
    assert not [t.name for t in threading.enumerate() if t.name.startswith("fsv-")]


def test_refused_sets_and_contigs_are_logged(caplog):
    """statuses the library returns are reported, and hard failures name their region"""
    import logging
    from focalsv_amd import _lib
    regions = [pipeline.RegionInput("chr21", 0, b"A", [], [], name="Region_chr21_S1_E2"), pipeline.RegionInput("chr21", 9, b"A", [], [], name="Region_chr21_S9_E99")]
    with caplog.at_level(logging.WARNING, logger="focalsv_amd"):
        failed = pipeline.report_statuses(regions, [0, 0, 1, 1], [1, 2, 1, 2], [0, 8 | 4, 0, _lib.EUNSUP], [0, 1, 1], [["contig_hp1_0"], ["contig_hp1_1"], ["contig_hp2_0"]],
                                          [0, _lib.EUNSUP, 1])
    assert failed == ["Region_chr21_S9_E99"]
    text = caplog.text
    assert "Region_chr21_S1_E2 hp2 read set: assembly status 12" in text and "insertion events dropped" in text and "no layout" in text
    assert "contig_hp1_1: the aligner refused this contig (status -6" in text and "contig_hp2_0: no chain" in text
    assert "Region_chr21_S9_E99 hp2 read set: assembly status -6" in text
