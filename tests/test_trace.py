"""ROCTx helper: no-op when off, balanced ranges when on, tolerant of a missing library."""


def test_trace_ranges_are_noops_when_disabled():
    from triton_client_amd.utils import trace

    trace.enable(False)
    assert not trace.enabled()
    assert trace.push("x") is False
    with trace.range("outer"):
        trace.mark("inside")


def test_trace_ranges_when_enabled():
    from triton_client_amd.utils import trace

    available = trace.enable(True)
    try:
        # with the library present ranges nest and unwind even through exceptions
        try:
            with trace.range("outer"):
                with trace.range("inner"):
                    trace.mark("m")
                raise RuntimeError("boom")
        except RuntimeError:
            pass
        assert trace.enabled() == available
        if available:
            assert trace.push("manual") is True
            trace.pop()
    finally:
        trace.enable(False)
