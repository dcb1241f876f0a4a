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
