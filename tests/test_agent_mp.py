"""RealtimeAgentMultiprocessing (realtime_codec_agent_amd/realtime_agent_mp.py; public surface of the reference's
realtime_agent_v2.py:791-928): spawned with fake resources on the CPU, with HIP resources and gpu_id=0 on the GPU."""
import numpy as np
import pytest

import mp_fakes
from agent_fakes import build_fakes, user_audio
from realtime_codec_agent_amd.realtime_agent_config import RealtimeAgentConfig
from realtime_codec_agent_amd.realtime_agent_mp import RealtimeAgentMultiprocessing, RealtimeAgentWorkerError
from realtime_codec_agent_amd.realtime_agent_v2 import RealtimeAgent

CFG = dict(chunk_size_secs=0.1, use_whisper=False, force_trans_after_inactivity_secs=0.0, force_response_after_inactivity_secs=0.0)


def test_worker_session_equals_in_process_session():
    """queue in -> chunk out, in order, identical to an agent run in this process; get_info, reset (pending frames are dropped),
    set_config_and_reset (chunk size changes), self-play item formats, clean shutdown."""
    audio = user_audio(16000)
    local = RealtimeAgent(resources=build_fakes()[0], config=RealtimeAgentConfig(**CFG))
    want = [local.process_audio(audio[s:s + 1600]) for s in range(0, 16000, 1600)]
    with RealtimeAgentMultiprocessing(config=RealtimeAgentConfig(**CFG), resources_factory=mp_fakes.fake_resources) as mpa:
        assert mpa.is_running() and mpa.next_output() is None
        info = mpa.get_info()
        assert info.chunk_size_samples == 1600 and info.sampling_rate == 16000 and info.total_secs == 0.0
        for s in range(0, 16000, 1600):
            mpa.queue_input(audio[s:s + 1600])
        got = [mpa.next_output(block=True) for _ in range(10)]
        assert all(np.array_equal(g[0], w) for g, w in zip(got, want))
        assert all(g[1] is None or g[1] > 0 for g in got)
        info = mpa.get_info()
        assert abs(info.total_secs - 1.0) < 1e-9 and info.sequence == local.get_sequence_str()
        assert np.array_equal(info.audio_history, local.get_audio_history())
        # reset while frames are queued: the old epoch's frames are skipped, its results never surface
        for s in range(0, 8000, 1600):
            mpa.queue_input(audio[s:s + 1600])
        mpa.reset()
        assert mpa.get_info().total_secs == 0.0 and mpa.next_output() is None
        mpa.queue_input(audio[:1600])
        assert np.array_equal(mpa.next_output(block=True)[0], want[0])     # a fresh session repeats the first frame
        # new config: 80 ms frames
        mpa.set_config_and_reset(RealtimeAgentConfig(**{**CFG, "chunk_size_secs": 0.08}))
        info = mpa.get_info()
        assert info.chunk_size_samples == 1280 and info.config.chunk_size_secs == 0.08
        mpa.queue_input((audio[:1280], None))                              # (chunk, ids | None) as a self-play partner sends it
        out, xrt = mpa.next_output(block=True)
        assert out.shape == (1280,)
        proc = mpa._process
    assert not proc.is_alive()


def test_self_play_mode_items():
    with RealtimeAgentMultiprocessing(config=RealtimeAgentConfig(**CFG), self_play_mode=True, resources_factory=mp_fakes.fake_resources) as a:
        a.queue_input(user_audio(1600))
        (chunk, ids), xrt = a.next_output(block=True)
        assert chunk.shape == (1600,) and len(ids) == 5
        a.queue_input((chunk, ids))                                         # ids given: the tokenizer is bypassed
        (chunk2, ids2), _ = a.next_output(block=True)
        assert len(ids2) == 5


def test_startup_failure_is_reported_not_spun_on():
    with pytest.raises(RealtimeAgentWorkerError) as e:
        RealtimeAgentMultiprocessing(config=RealtimeAgentConfig(**CFG), resources_factory=mp_fakes.broken_resources)
    assert "no such model file" in str(e.value)
    lazy = RealtimeAgentMultiprocessing(wait_until_running=False, config=RealtimeAgentConfig(**CFG), resources_factory=mp_fakes.broken_resources)
    with pytest.raises(RealtimeAgentWorkerError):
        lazy.wait_until_running()
    assert not lazy.is_running()
    lazy.close()


def test_frame_failure_surfaces_and_the_session_survives():
    with RealtimeAgentMultiprocessing(config=RealtimeAgentConfig(**CFG), resources_factory=mp_fakes.flaky_resources) as a:
        audio = user_audio(4800)
        for s in range(0, 4800, 1600):
            a.queue_input(audio[s:s + 1600])
        assert a.next_output(block=True)[0].shape == (1600,)
        with pytest.raises(RealtimeAgentWorkerError) as e:
            a.next_output(block=True)
        assert "synthetic LM failure" in str(e.value)
        assert a.is_running()
        a.reset()
        a.queue_input(audio[:1600])
        assert a.next_output(block=True)[0].shape == (1600,)


@pytest.mark.gpu
def test_gpu_worker_pinned_to_gpu_0():
    """gpu_id=0: the worker sets HIP_VISIBLE_DEVICES before loading, builds HIP resources, and produces the same frames as an
    in-process agent over the same seeds."""
    audio = user_audio(1280 * 12)
    cfg = RealtimeAgentConfig(**{**CFG, "chunk_size_secs": 0.08})
    local = RealtimeAgent(resources=mp_fakes.tiny_gpu_resources(), config=cfg)
    want = [local.process_audio(audio[s:s + 1280]) for s in range(0, len(audio), 1280)]
    with RealtimeAgentMultiprocessing(config=cfg, gpu_id=0, resources_factory=mp_fakes.tiny_gpu_resources) as a:
        for s in range(0, len(audio), 1280):
            a.queue_input(audio[s:s + 1280])
        got = [a.next_output(block=True) for _ in range(12)]
        assert all(np.array_equal(g[0], w) for g, w in zip(got, want))
        info = a.get_info()
        assert info.sequence == local.get_sequence_str() and abs(info.total_secs - 0.96) < 1e-9
