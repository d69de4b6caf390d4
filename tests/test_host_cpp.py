"""The C++ block-graph mirror (qdsp_amd/host/dsp): CPU checks of the parts that need no GPU
(window designers vs the oracle's restatement of src/dsp/window.h, stream/block protocol),
and -- marked gpu -- whole source -> block -> sink graphs against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as O
from conftest import rel_rms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "qdsp_amd", "host")
BIN = os.path.join(HOST, "build", "graph_check")


@pytest.fixture(scope="module")
def harness():
    if not all(os.path.exists(os.path.join(os.path.dirname(BIN), b)) for b in ("graph_check", "graph_check_big", "patch_b_fir")):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL, timeout=300)
    return BIN


def run(args, timeout=180):
    return subprocess.run(args, check=True, timeout=timeout, capture_output=True, text=True)


def test_window_designers_match_oracle_bit_for_bit(harness, tmp_path):
    out = tmp_path / "taps.f32"
    run([harness, "taps", str(out)])
    raw = np.fromfile(out, dtype=np.float32)
    tables, i = [], 0
    while i < len(raw):
        n = int(raw[i])
        tables.append(raw[i + 1:i + 1 + n])
        i += 1 + n
    assert [len(t) for t in tables] == [63, 63, 31, 97, 63]
    assert np.array_equal(tables[0], O.blackman_taps(0.1, 1.0, 63))
    assert np.array_equal(tables[1], O.blackman_bandpass_taps(0.05, 0.2, 1.0, 63))
    assert np.array_equal(tables[2], O.rrc_taps(31, 4.0, 1.0, 0.35))
    L, M, vt = O.vfo_design(2.4e6, 240e3, 200e3)
    assert (L, M) == (1, 10) and np.array_equal(tables[3], vt)
    assert np.array_equal(tables[4], O.blackman_taps(0.1, 1.0, 63, factor=3.0))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "vectors.npz"))
    assert np.array_equal(tables[0], gold["taps63"]) and np.array_equal(tables[3], gold["vfo_taps"])


def test_stream_and_block_protocol(harness):
    r = run([harness, "stream"], timeout=60)
    assert "self-test ok" in r.stdout


def test_headers_keep_the_reference_surface():
    """Names a graph written against the reference uses must exist in the mirror."""
    need = {
        "dsp/stream.h": ["untyped_steam", "class stream", "bool swap(int size)", "int read()", "void flush()", "stopWriter", "clearWriteStop",
                         "stopReader", "clearReadStop", "T* writeBuf", "T* readBuf", "STREAM_BUFFER_SIZE 1000000"],
        "dsp/block.h": ["generic_unnamed_block", "class generic_block", "generic_hier_block", "registerInput", "registerOutput", "unregisterInput",
                        "tempStart", "tempStop", "ctrlMtx", "friend BLOCK", "calcOutSize"],
        "dsp/filter.h": ["class FIR", "updateWindow", "setInput", "stream<T> out"],
        "dsp/resampling.h": ["class PolyphaseResampler", "setInSampleRate", "setOutSampleRate", "getInterpolation", "getDecimation", "updateWindow",
                             "calcOutSize", "stream<T> out"],
        "dsp/processing.h": ["class FrequencyXlator", "setInputSize", "setSampleRate", "getSampleRate", "setFrequency", "getFrequency"],
        "dsp/vfo.h": ["class VFO", "setInSampleRate", "setOutSampleRate(float outSampleRate, float bandWidth)", "setOffset", "setBandwidth",
                      "stream<complex_t>* out"],
        "dsp/window.h": ["generic_window", "BlackmanWindow", "BlackmanBandpassWindow", "RRCTaps", "getTapCount", "createTaps"],
        "dsp/types.h": ["struct complex_t", "struct stereo_t", "FL_M_PI 3.1415926535f", "fastPhase", "fastAmplitude", "conj()"],
        "dsp/routing.h": ["class Splitter", "bindStream", "unbindStream", "setInput"],
        "dsp/math.h": ["class Add", "class Substract", "class Multiply", "stream<T> out", "a_count != b_count"],
        "dsp/source.h": ["class SineSource", "setBlockSize", "getBlockSize", "setFrequency", "class HandlerSource", "setHandler"],
        "dsp/sink.h": ["class HandlerSink", "class NullSink", "class FileSink"],
        "wav.h": ["class WavWriter", "writeSamples"], "wavreader.h": ["class WavReader", "readSamples", "getSampleRate", "isValid"],
    }
    for f, names in need.items():
        txt = open(os.path.join(HOST, f)).read()
        for n in names:
            assert n in txt, (f, n)


# ------------------------------------------------------------------------------------ GPU graphs
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("graph")
    x = O.synth_iq(0, 300_000, seed=99)
    x.tofile(d / "x.cf32")
    np.ascontiguousarray(x.real).tofile(d / "x.f32")
    return d, x


def blocks(op, x, b):
    return np.concatenate([op.process(x[i:i + b]) for i in range(0, len(x), b)])


def test_block_failures_are_counted_where_a_host_can_poll_them(harness):
    """A block whose GPU call fails reports once on stderr and ends its worker (the reference has no error channel, block.h:55-57); the
    failure is also counted process-wide: dsp::hipBlockErrors() / hipBlockLastError().  Runs with or without a GPU: an empty tap table
    (or, here, no device at all) cannot make a handle."""
    r = subprocess.run([harness, "fail"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors 1" in r.stdout and "FIR::init" in r.stdout and "[qdsp_hip] FIR::init" in r.stderr


@gpu
def test_graph_fir63(harness, data):
    d, x = data
    run([harness, "fir63", str(d / "x.cf32"), str(d / "y.cf32"), "65536"])
    y = np.fromfile(d / "y.cf32", dtype=np.complex64)
    want = blocks(O.Fir(O.blackman_taps(0.1, 1.0, 63)), x, 65536)
    assert len(y) == len(want) and rel_rms(y, want) < 2e-6


@gpu
def test_graph_fir256_custom_window_and_float(harness, data):
    d, x = data
    taps = O.lowpass_taps_f64(256, 1 / 16)
    taps.tofile(d / "t256.f32")
    run([harness, "fir", str(d / "x.cf32"), str(d / "y.cf32"), "100000", str(d / "t256.f32")])
    y = np.fromfile(d / "y.cf32", dtype=np.complex64)
    assert rel_rms(y, O.Fir(taps).process(x)) < 2e-6          # 100000-sample blocks -> FFT path
    run([harness, "fir", str(d / "x.cf32"), str(d / "y2.cf32"), "4096", str(d / "t256.f32")])
    y2 = np.fromfile(d / "y2.cf32", dtype=np.complex64)
    assert np.array_equal(y2, O.Fir(taps, acc=O.ACC_FMA).process(x))  # small blocks -> direct form, bit-exact
    # FIR<float>: blocks of under 2^19 tap-samples (1024 samples at 256 taps) stay on the direct form (bit-exact); 50000-sample
    # blocks go to the one-wave overlap-save kernel (two real segments per transform: 6.5 us against ~50 for the direct
    # kernel): tolerance, not equality
    run([harness, "firf", str(d / "x.f32"), str(d / "yf.f32"), "1024", str(d / "t256.f32")])
    yf = np.fromfile(d / "yf.f32", dtype=np.float32)
    xr = np.ascontiguousarray(x.real)
    assert np.array_equal(yf, O.Fir(taps, complex_data=False, acc=O.ACC_FMA).process(xr))
    run([harness, "firf", str(d / "x.f32"), str(d / "yf2.f32"), "50000", str(d / "t256.f32")])
    yf2 = np.fromfile(d / "yf2.f32", dtype=np.float32)
    assert rel_rms(yf2, O.Fir(taps, complex_data=False, acc=O.ACC_F64).process(xr)) < 2e-6


@gpu
def test_graph_resampler_xlator_vfo(harness, data):
    d, x = data
    b = 50_000
    # PolyphaseResampler(BlackmanWindow(cutoff 12k, trans 6k, 48k), 48k -> 32k): L=2, M=3
    run([harness, "resamp", str(d / "x.cf32"), str(d / "yr.cf32"), str(b), "48000", "32000", "12000", "6000"])
    yr = np.fromfile(d / "yr.cf32", dtype=np.complex64)
    L, M = O.resamp_ratio(48000.0, 32000.0)
    n = O.blackman_tap_count(12000.0, 6000.0, 48000.0)
    taps = O.blackman_taps(12000.0, 48000.0, n, factor=float(L))
    want = blocks(O.Resampler(taps, L, M), x, b)
    assert (L, M) == (2, 3) and len(yr) == len(want) and rel_rms(yr, want) < 2e-6
    # FrequencyXlator(2.4 MHz, 123456 Hz)
    run([harness, "xlate", str(d / "x.cf32"), str(d / "yx.cf32"), "4096", "2400000", "123456"])
    yx = np.fromfile(d / "yx.cf32", dtype=np.complex64)
    wx = blocks(O.Xlator(2.4e6, 123456.0, exact=True, volk_gain=True), x, 4096)
    assert np.abs(yx - wx).max() < 6e-7
    # VFO(offset 300k, 2.4M -> 240k, bw 200k)
    run([harness, "vfo", str(d / "x.cf32"), str(d / "yv.cf32"), str(b), "300000", "2400000", "240000", "200000"])
    yv = np.fromfile(d / "yv.cf32", dtype=np.complex64)
    wv = blocks(O.Vfo(300e3, 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True), x, b)
    assert len(yv) == len(wv) and rel_rms(yv, wv) < 2e-6


@gpu
def test_graph_callers_filter_setups(harness, data):
    """The filter set-ups of the reference's own callers, through the GPU graph (missing #3 / #4):
    * FIR<complex_t> fed by RRCTaps (PSKDemod's matched filter, demodulator.h:586-587) -- with an EVEN requested tap count, where
      RRCTaps::createTaps does `tapCount |= 1` (window.h:183) and designs tapCount + 1 taps, normalised over all of them, of which
      the FIR (tapCount = getTapCount(), filter.h:24-26) uses the first tapCount; the odd case beside it;
    * FIR<float> fed by BlackmanBandpassWindow (StereoFMDemod's 19 kHz pilot filter, demodulator.h:216-217);
    * PolyphaseResampler<stereo_t> (resampling.h:120: the complex kernel on a float pair)."""
    d, x = data
    for n_req in (32, 31):
        run([harness, "firrrc", str(d / "x.cf32"), str(d / "yrrc.cf32"), "4096", str(n_req), "4", "1", "0.35"])
        y = np.fromfile(d / "yrrc.cf32", dtype=np.complex64)
        taps = O.rrc_taps(n_req | 1, 4.0, 1.0, 0.35)[:n_req]       # what the reference's createTaps leaves in taps[0 .. n_req)
        assert len(taps) == n_req
        assert np.array_equal(y, blocks(O.Fir(taps, acc=O.ACC_FMA), x, 4096)), n_req       # small blocks: direct form, bit-exact
        run([harness, "firrrc", str(d / "x.cf32"), str(d / "yrrc2.cf32"), "100000", str(n_req), "4", "1", "0.35"])
        y2 = np.fromfile(d / "yrrc2.cf32", dtype=np.complex64)
        assert rel_rms(y2, O.Fir(taps, acc=O.ACC_F64).process(x)) < 2e-6, n_req
    # pilot filter: BlackmanBandpassWindow(cutoff 300, trans 100, offset 19000, 48000) on a real stream
    xr = np.ascontiguousarray(x.real)
    nbp = O.blackman_tap_count(300.0, 100.0, 48000.0)
    tbp = O.blackman_bandpass_taps(300.0, 19000.0, 48000.0, nbp)
    for blk in (1000, 50000):
        run([harness, "firbp", str(d / "x.f32"), str(d / "ybp.f32"), str(blk), "300", "100", "19000", "48000"])
        ybp = np.fromfile(d / "ybp.f32", dtype=np.float32)
        want = blocks(O.Fir(tbp, complex_data=False, acc=O.ACC_F64), xr, blk)
        # (1919 taps: a k-ordered FP32 sum of that length sits near 1e-6 of the FP64 value by itself)
        assert nbp == 1919 and len(ybp) == len(want) and rel_rms(ybp, want) < 5e-6, (blk, rel_rms(ybp, want))
    # PolyphaseResampler<stereo_t>(BlackmanWindow(12k, 6k, 48k), 48k -> 32k): (l, r) pairs == (re, im) pairs
    b = 50_000
    run([harness, "resampst", str(d / "x.cf32"), str(d / "yst.cf32"), str(b), "48000", "32000", "12000", "6000"])
    yst = np.fromfile(d / "yst.cf32", dtype=np.complex64)
    L, M = O.resamp_ratio(48000.0, 32000.0)
    n = O.blackman_tap_count(12000.0, 6000.0, 48000.0)
    want = blocks(O.Resampler(O.blackman_taps(12000.0, 48000.0, n, factor=float(L)), L, M), x, b)
    assert len(yst) == len(want) and rel_rms(yst, want) < 2e-6
    run([harness, "resamp", str(d / "x.cf32"), str(d / "ycx.cf32"), str(b), "48000", "32000", "12000", "6000"])
    assert np.array_equal(yst, np.fromfile(d / "ycx.cf32", dtype=np.complex64))      # same kernel, same bits


@gpu
def test_graph_with_larger_stream_buffers(harness, tmp_path):
    """STREAM_BUFFER_SIZE is a build-time choice of the graph (dsp/stream.h; the reference's 1e6, src/dsp/stream.h:7, is the default):
    the same harness compiled with -DSTREAM_BUFFER_SIZE=16777216 hands 4 000 000-sample blocks to the library -- one call of the
    chip-filling kernels instead of four latency-bound ones -- and the results still follow the oracle fed the same blocks."""
    big = os.path.join(os.path.dirname(harness), "graph_check_big")
    n, b = 9_000_000, 4_000_000
    x = O.synth_iq(0, n, seed=424)
    x.tofile(tmp_path / "x.cf32")
    taps = O.lowpass_taps_f64(256, 1 / 16)
    taps.tofile(tmp_path / "t256.f32")
    run([big, "fir", str(tmp_path / "x.cf32"), str(tmp_path / "y.cf32"), str(b), str(tmp_path / "t256.f32")], timeout=600)
    y = np.fromfile(tmp_path / "y.cf32", dtype=np.complex64)
    want = O.Fir(taps).process(x)                                    # (an FIR's output does not depend on the block cut)
    assert len(y) == n and rel_rms(y, want) < 2e-6
    # VFO(offset 300k, 2.4M -> 240k, bw 200k): per-block phase restart of the resampler, NCO carried across the blocks
    run([big, "vfo", str(tmp_path / "x.cf32"), str(tmp_path / "yv.cf32"), str(b), "300000", "2400000", "240000", "200000"], timeout=600)
    yv = np.fromfile(tmp_path / "yv.cf32", dtype=np.complex64)
    wv = blocks(O.Vfo(300e3, 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True), x, b)
    assert len(yv) == len(wv) and rel_rms(yv, wv) < 2e-6
    # the default build refuses such a block (it does not fit its streams)
    r = subprocess.run([harness, "fir", str(tmp_path / "x.cf32"), str(tmp_path / "y2.cf32"), str(b), str(tmp_path / "t256.f32")], capture_output=True, text=True)
    assert r.returncode != 0


@gpu
def test_graph_wav_config1(harness, tmp_path):
    """BASELINE configs[0]: 16-bit stereo (I,Q) WAV -> 63-tap lowpass FIR (SURVEY 8d config 1:
    two tones + uniform noise, mt19937(1234), 2.4 Msps, 2^20 frames, blocks of 65536)."""
    fs, n = 2_400_000, 1 << 20
    rng = np.random.Generator(np.random.MT19937(1234))
    t = np.arange(n) / fs
    sig = 0.35 * np.exp(2j * np.pi * 100e3 * t) + 0.25 * np.exp(-2j * np.pi * 700e3 * t) + 0.1 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))
    pcm = np.empty(2 * n, dtype=np.int16)
    pcm[0::2] = np.clip(np.round(sig.real * 32767), -32768, 32767)
    pcm[1::2] = np.clip(np.round(sig.imag * 32767), -32768, 32767)
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", pcm.nbytes + 36, b"WAVE", b"fmt ", 16, 1, 2, fs, fs * 4, 4, 16, b"data", pcm.nbytes)
    wav = tmp_path / "iq.wav"
    wav.write_bytes(hdr + pcm.tobytes())
    run([harness, "wavfir", str(wav), str(tmp_path / "y.cf32"), "65536"])
    y = np.fromfile(tmp_path / "y.cf32", dtype=np.complex64)
    iq = (pcm[0::2].astype(np.float32) / np.float32(32768.0) + 1j * (pcm[1::2].astype(np.float32) / np.float32(32768.0))).astype(np.complex64)
    ntaps = O.blackman_tap_count(0.1 * fs, 4.0 * fs / 63.0, float(fs))
    taps = O.blackman_taps(0.1 * fs, float(fs), ntaps)
    want = blocks(O.Fir(taps), iq, 65536)
    assert ntaps == 63 and len(y) == n and rel_rms(y, want) < 2e-6
    # the 700 kHz tone is outside the 240 kHz passband, the 100 kHz tone inside
    spec = np.abs(np.fft.fft(y[4096:4096 + 65536] * np.hanning(65536)))
    f = np.fft.fftfreq(65536, 1 / fs)
    assert spec[np.argmin(np.abs(f - 100e3))] > 30 * spec[np.argmin(np.abs(f + 700e3))]


@gpu
def test_graph_device_resident_chain(harness, data):
    """source -> FrequencyXlator -> FIR -> PolyphaseResampler -> sink: the two inner links stay on
    the device (stream.h device-resident companion, *_process_ex); result == the same blocks
    chained through the oracle."""
    d, x = data
    taps = O.lowpass_taps_f64(128, 0.2)
    taps.tofile(d / "t128.f32")
    b = 40_000
    run([harness, "chain", str(d / "x.cf32"), str(d / "yc.cf32"), str(b), str(d / "t128.f32"), "48000", "-5000", "48000", "12000"])
    y = np.fromfile(d / "yc.cf32", dtype=np.complex64)
    xl = O.Xlator(48000.0, -5000.0, exact=True, volk_gain=True)
    fir = O.Fir(taps, acc=O.ACC_F64)
    L, M = O.resamp_ratio(48000.0, 12000.0)
    n = O.blackman_tap_count(6000.0, 6000.0, 48000.0)
    rs = O.Resampler(O.blackman_taps(6000.0, 48000.0, n, factor=float(L)), L, M, acc=O.ACC_F64)
    want = np.concatenate([rs.process(fir.process(xl.process(x[i:i + b]))) for i in range(0, len(x), b)])
    assert (L, M) == (1, 4) and len(y) == len(want) and rel_rms(y, want) < 3e-6


@gpu
def test_graph_many_small_blocks_through_pipelined_links(harness, data):
    """The same three-block chain and the Splitter fan-out fed hundreds of small blocks back to back: every
    hand-over between the GPU blocks is a pipelined link (nobody waits for a kernel), the last block leaves the
    wait to the sink's read().  A missed ordering -- a buffer read before it was written, or reused before it was
    read -- shows up as a wrong block."""
    d, x = data
    taps = O.lowpass_taps_f64(128, 0.2)
    taps.tofile(d / "t128b.f32")
    b = 1_000
    run([harness, "chain", str(d / "x.cf32"), str(d / "ycs.cf32"), str(b), str(d / "t128b.f32"), "48000", "-5000", "48000", "12000"])
    y = np.fromfile(d / "ycs.cf32", dtype=np.complex64)
    xl = O.Xlator(48000.0, -5000.0, exact=True, volk_gain=True)
    fir = O.Fir(taps, acc=O.ACC_F64)
    L, M = O.resamp_ratio(48000.0, 12000.0)
    n = O.blackman_tap_count(6000.0, 6000.0, 48000.0)
    rs = O.Resampler(O.blackman_taps(6000.0, 48000.0, n, factor=float(L)), L, M, acc=O.ACC_F64)
    want = np.concatenate([rs.process(fir.process(xl.process(x[i:i + b]))) for i in range(0, len(x), b)])
    assert len(x) // b >= 100 and len(y) == len(want) and rel_rms(y, want) < 3e-6
    nv = 4
    b = 2_000
    run([harness, "split", str(d / "x.cf32"), str(d / "yss"), str(b), str(nv), "2400000", "240000", "200000"])
    for i in range(nv):
        y = np.fromfile(str(d / "yss") + f".{i}.cf32", dtype=np.complex64)
        off = np.float32((np.float32(i) - np.float32(nv - 1) / np.float32(2.0)) * np.float32(2.4e6) / np.float32(nv))
        v = O.Vfo(float(off), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
        want = np.concatenate([v.process(x[j:j + b]) for j in range(0, len(x), b)])
        assert len(y) == len(want) and rel_rms(y, want) < 3e-6, i


@gpu
def test_graph_splitter_to_vfos(harness, data):
    """source -> Splitter -> 4 x VFO -> sinks (the channelizer shape of the reference)."""
    d, x = data
    b, n = 50_000, 4
    run([harness, "split", str(d / "x.cf32"), str(d / "ys"), str(b), str(n), "2400000", "240000", "200000"])
    for i in range(n):
        y = np.fromfile(str(d / "ys") + f".{i}.cf32", dtype=np.complex64)
        off = np.float32((np.float32(i) - np.float32(n - 1) / np.float32(2.0)) * np.float32(2.4e6) / np.float32(n))
        v = O.Vfo(float(off), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
        want = np.concatenate([v.process(x[j:j + b]) for j in range(0, len(x), b)])
        assert len(y) == len(want) and rel_rms(y, want) < 3e-6, i


@gpu
def test_graph_bank_retune_and_reconfigure(harness, data):
    """Splitter -> 4 x VFO runs as a bank (one batched launch per block, dsp/vfo_bank.h).  (1) setOffset on one VFO while
    the graph is live reaches the bank's channel: phase continuous, increment changed from block K on -- exact against the
    oracle.  (2) setBandwidth on one VFO takes the bank down: every VFO gets its channel's NCO phase and filter history back
    (qdsp_hip_chan_cf32_move_channel_state) and runs its own kernel again: the untouched channels stay exact against the
    oracle across the hand-back, the re-designed one up to the change."""
    d, x = data
    b, n, K = 20_000, 4, 3
    offs = [np.float32((np.float32(i) - np.float32(n - 1) / np.float32(2.0)) * np.float32(2.4e6) / np.float32(n)) for i in range(n)]
    nblk = (len(x) + b - 1) // b
    assert nblk > K + 2
    new_off = 123456.0
    run([harness, "splitretune", str(d / "x.cf32"), str(d / "yrt"), str(b), str(n), "2400000", "240000", "200000", str(K), str(new_off)])
    for i in range(n):
        y = np.fromfile(str(d / "yrt") + f".{i}.cf32", dtype=np.complex64)
        v = O.Vfo(float(offs[i]), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
        want = []
        for k, j in enumerate(range(0, len(x), b)):
            if i == 1 and k == K:
                v.xl.delta[:] = O.Xlator(2.4e6, -new_off).delta      # VFO: xlator(-offset), vfo.h:28
            want.append(v.process(x[j:j + b]))
        want = np.concatenate(want)
        assert len(y) == len(want) and rel_rms(y, want) < 3e-6, i
    run([harness, "splitretune", str(d / "x.cf32"), str(d / "yrc"), str(b), str(n), "2400000", "240000", "200000", str(K), "0", "reconf"])
    per = b // 10
    for i in range(n):
        y = np.fromfile(str(d / "yrc") + f".{i}.cf32", dtype=np.complex64)
        v = O.Vfo(float(offs[i]), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
        want = np.concatenate([v.process(x[j:j + b]) for j in range(0, len(x), b)])
        assert len(y) == len(want)
        assert rel_rms(y[:K * per], want[:K * per]) < 3e-6, i          # banked blocks
        if i != 0:                                                     # own kernel from block K on, state handed back: no glitch
            assert rel_rms(y, want) < 3e-6, i


@gpu
def test_graph_shard_over_the_c_ring(harness, data):
    """graph_check shard: the time-sharded path driven from C++ through the C ABI (include/qdsp_hip.h "ring"): one rank as its own
    ring neighbour filters the stream chunk by chunk, every chunk's history delivered by qdsp_hip_ring_post / _complete (RCCL
    send + recv on the ring's stream, posted one step ahead) and installed with qdsp_hip_fir_cf32_set_history_dev: the
    concatenated chunks equal FIR<complex_t> over the whole stream (src/dsp/filter.h:51-74 carries exactly those ntaps - 1
    samples from one run() to the next)."""
    d, x = data
    taps = O.lowpass_taps_f64(256, 1 / 16)
    taps.tofile(d / "t256s.f32")
    n = 60_000
    run([harness, "shard", str(d / "x.cf32"), str(d / "ysh.cf32"), str(n), str(d / "t256s.f32")])
    y = np.fromfile(d / "ysh.cf32", dtype=np.complex64)
    want = O.Fir(taps, acc=O.ACC_F64).process(x)[:len(x) // n * n]
    assert len(y) == len(want) and rel_rms(y, want) < 2e-6
    for k in range(1, len(x) // n):          # the first outputs of every chunk: a wrong halo is an O(1) error there
        assert rel_rms(y[k * n:k * n + 64], want[k * n:k * n + 64]) < 1e-5, k


@gpu
def test_graph_bank_rebind_on_a_live_splitter(harness, data):
    """bindStream / unbindStream on a Splitter whose outputs run as a bank (how VFOs are added to a running graph,
    src/dsp/routing.h:27-45): the bank is taken down while token blocks of the last banked block may still be waiting in the
    links.  A token carries its mark with it (stream<T>::readIsToken), so every VFO passes the bank's output on instead of
    filtering a never-written buffer and advancing its NCO twice: the four original channels stay exact against the oracle
    across both re-plumbings, the leg that was bound at block K produces blocks K .. K+2 of a VFO started there."""
    d, x = data
    b, n, K = 20_000, 4, 3
    offs = [np.float32((np.float32(i) - np.float32(n - 1) / np.float32(2.0)) * np.float32(2.4e6) / np.float32(n)) for i in range(n)]
    run([harness, "splitretune", str(d / "x.cf32"), str(d / "yrb"), str(b), str(n), "2400000", "240000", "200000", str(K), "0", "rebind"])
    for i in range(n):
        y = np.fromfile(str(d / "yrb") + f".{i}.cf32", dtype=np.complex64)
        v = O.Vfo(float(offs[i]), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
        want = np.concatenate([v.process(x[j:j + b]) for j in range(0, len(x), b)])
        assert len(y) == len(want) and rel_rms(y, want) < 3e-6, i
    y = np.fromfile(str(d / "yrb") + f".{n}.cf32", dtype=np.complex64)
    v = O.Vfo(float(np.float32(0.125) * np.float32(2.4e6)), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
    want = np.concatenate([v.process(x[j:j + b]) for j in range(K * b, (K + 3) * b, b)])
    assert len(y) == len(want) and rel_rms(y, want) < 3e-6


@gpu
def test_graph_multiply_into_splitter_to_vfos(harness, data):
    """source -> Splitter -> Multiply(x, x) -> Splitter -> 8 x VFO -> sinks.  The second Splitter's input is a
    device-resident block from a producer that does not launch into the library's pipelined stream (the math block has
    its own stream), while its outputs are pipelined links: its queued device-to-device copies must have read the block
    before it flushes the input (qdsp_hip_memcpy_d2d_link), or the Multiply overwrites it two blocks later."""
    d, x = data
    b, n = 25_000, 8
    run([harness, "mulsplit", str(d / "x.cf32"), str(d / "yms"), str(b), str(n), "2400000", "240000", "200000"])
    x2 = O.math_op(2, x, x)
    for i in range(n):
        y = np.fromfile(str(d / "yms") + f".{i}.cf32", dtype=np.complex64)
        off = np.float32((np.float32(i) - np.float32(n - 1) / np.float32(2.0)) * np.float32(2.4e6) / np.float32(n))
        v = O.Vfo(float(off), 2.4e6, 240e3, 200e3, exact_nco=True, volk_gain=True)
        want = np.concatenate([v.process(x2[j:j + b]) for j in range(0, len(x2), b)])
        assert len(y) == len(want) and rel_rms(y, want) < 3e-6, i


@gpu
def test_integration_patch_b_fir(harness, data):
    """INTEGRATION.md section B compiled: a block that keeps its own stream / window types and only swaps the
    VOLK loop for qdsp_hip_fir_cf32_process (examples/patch_b_fir.cpp)."""
    d, x = data
    exe = os.path.join(os.path.dirname(harness), "patch_b_fir")
    assert os.path.exists(exe)
    taps = O.lowpass_taps_f64(256, 0.0625).astype(np.float32)
    taps.tofile(d / "t256b.f32")
    run([exe, str(d / "x.cf32"), str(d / "yb.cf32"), "65536", str(d / "t256b.f32")])
    y = np.fromfile(d / "yb.cf32", dtype=np.complex64)
    want = blocks(O.Fir(taps, acc=O.ACC_F64), x, 65536)
    assert len(y) == len(want) and rel_rms(y, want) < 2e-6


@gpu
@pytest.mark.parametrize("op", ["add", "sub", "mul"])
def test_graph_math_blocks(harness, data, op):
    """source -> Splitter -> { FrequencyXlator, identity } -> Add | Substract | Multiply -> sink
    (src/dsp/math.h): both inputs of the math block arrive over links, one of them device-resident."""
    d, x = data
    b = 30_000
    run([harness, "math", str(d / "x.cf32"), str(d / f"ym_{op}.cf32"), str(b), op, "48000", "1234"])
    y = np.fromfile(d / f"ym_{op}.cf32", dtype=np.complex64)
    xl = O.Xlator(48000.0, 1234.0, exact=True, volk_gain=True)
    a = np.concatenate([xl.process(x[i:i + b]) for i in range(0, len(x), b)])
    want = O.math_op({"add": 0, "sub": 1, "mul": 2}[op], a, x)
    assert len(y) == len(want) and rel_rms(y, want) < 2e-6


@gpu
def test_graph_live_bench_modes(harness):
    """graph_check bench: a free-running SineSource -> VFO -> sink graph (and the same two blocks unfused with a
    device-resident link) through the block API; the harness prints the rate DESIGN.md quotes."""
    import re
    import subprocess

    for kind in ("vfo", "chain"):
        out = subprocess.run([harness, "bench", kind, "65536", "40", "2400000", "48000"], check=True, capture_output=True,
                             text=True, timeout=120).stdout
        m = re.search(r"= ([0-9.]+) Msamples/s in, ([0-9.]+) us per block", out)
        assert m and float(m.group(1)) > 10.0, out


@gpu
def test_graph_sine_source(harness, tmp_path):
    """SineSource (device NCO) alone, and feeding a FIR through a device-resident link."""
    bs, nb, fs, f = 4096, 8, 48000.0, 1234.0
    run([harness, "sine", str(tmp_path / "s.cf32"), str(bs), str(nb), str(fs), str(f)])
    y = np.fromfile(tmp_path / "s.cf32", dtype=np.complex64)
    ones = np.ones(bs * nb, np.complex64)
    xl = O.Xlator(fs, f, exact=True, volk_gain=True)      # rotator over ones, one call per block
    want = np.concatenate([xl.process(ones[i:i + bs]) for i in range(0, len(ones), bs)])
    assert len(y) == len(want) and np.abs(y - want).max() < 6e-7
    g = O.Xlator(fs, f)                                    # the reference's recursive phasor
    wg = np.concatenate([g.process(ones[i:i + bs]) for i in range(0, len(ones), bs)])
    assert rel_rms(y, wg) < 1e-5
    taps = O.lowpass_taps_f64(64, 0.1)
    taps.tofile(tmp_path / "t.f32")
    run([harness, "sine", str(tmp_path / "sf.cf32"), str(bs), str(nb), str(fs), str(f), str(tmp_path / "t.f32")])
    yf = np.fromfile(tmp_path / "sf.cf32", dtype=np.complex64)
    fir = O.Fir(taps, acc=O.ACC_F64)
    wf = np.concatenate([fir.process(want[i:i + bs]) for i in range(0, len(want), bs)])
    assert rel_rms(yf, wf) < 2e-6
