"""Python mirror of the reference's interface for the logo / CM / KFM analysis path, on top of the C ABI.

Class names, argument meaning and error behaviour follow the reference (Amatsukaze/LogoScan.hpp):
``LogoFrame(ctx, logofiles, maskratio)`` + ``scanFrames`` / ``selectLogo`` / ``writeResult`` /
``getBestLogo`` / ``getLogoRatio``; ``AMTAnalyzeLogo(clip, logopath, maskratio)``;
``AMTEraseLogo(clip, analyzeclip, logopath, logofpath, mode, maxfade)``; ``ScanLogo(...)``.
A "clip" here is a :class:`DeviceClip`: planar 4:2:0 frames resident in HBM as torch tensors
(torch is plumbing for device memory and streams only).  Failures raise :class:`AmtError` with the
message the C ABI keeps on its context -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import binding


class AmtError(RuntimeError):
    pass


def _p(t):
    """device pointer of a torch tensor / host pointer of a numpy array / None"""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data_as(C.c_void_p)
    return C.c_void_p(t.data_ptr())


class Context:
    """AMTContext stand-in bound to one GPU (StreamUtils.hpp:343-511)."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        if use_torch_stream:
            # torch BEFORE the library: both link libamdhip64, and the process must end up with one HIP runtime.  With torch's copy
            # loaded first the library binds to it; the other way round torch initialises a second runtime and reports
            # "No HIP GPUs are available".
            import torch
            torch.cuda.init()
        self.lib = binding.load()
        buf = C.create_string_buffer(2048)
        if self.lib.amtgpu_hip_runtimes_loaded(buf, len(buf)) > 1:
            raise AmtError("more than one HIP runtime is mapped into this process (" + ", ".join(buf.value.decode().split()) + "): device "
                           "memory and streams of one are unknown to the other -- load torch (or whichever component brings its own "
                           "libamdhip64) before amatsukaze_amd")
        self.h = self.lib.amtgpu_context_create(device)
        if not self.h:
            raise AmtError(f"amtgpu_context_create({device}) failed: no usable HIP device")
        self.device = device
        if use_torch_stream:
            # Launch on torch's current stream so that kernels are stream-ordered with torch fills / copies / reads of the
            # same tensors.  torch's default stream is the legacy null stream (handle 0), which the ABI spells
            # AMTGPU_STREAM_LEGACY_DEFAULT (NULL would select the context's own non-blocking stream).
            import torch
            hs = int(torch.cuda.current_stream(device).cuda_stream)
            self.lib.amtgpu_context_set_stream(self.h, C.c_void_p(hs if hs else 1))

    def cu_count(self) -> int:
        return self.lib.amtgpu_device_cu_count(self.h)

    def use_cu_range(self, first_cu: int, num_cus: int):
        """Launch this context's kernels on compute units [first_cu, first_cu + num_cus) only (amtgpu_stream_create_cu_range): one
        context per partition lets bandwidth-bound and arithmetic-bound passes run beside each other.  Returns the stream as a
        torch.cuda.ExternalStream so that the caller can order it against other streams (wait_stream / record_event)."""
        import torch
        st = self.lib.amtgpu_stream_create_cu_range(self.h, first_cu, num_cus)
        self.check(st, "stream_create_cu_range")
        self.check(self.lib.amtgpu_context_set_stream(self.h, C.c_void_p(st)))
        self._cu_stream = st
        return torch.cuda.ExternalStream(st, device=self.device)

    def check(self, ok, what=""):
        if not ok:
            raise AmtError((what + ": " if what else "") + self.lib.amtgpu_last_error(self.h).decode(errors="replace"))
        return ok

    def synchronize(self):
        self.check(self.lib.amtgpu_context_synchronize(self.h))

    def profile(self, on: bool = True):
        self.check(self.lib.amtgpu_profile_enable(self.h, 1 if on else 0))

    def profile_report(self):
        """{kernel: (calls, total_ms)} measured with HIP events on the launch stream"""
        buf = C.create_string_buffer(1 << 14)
        n = self.lib.amtgpu_profile_report(self.h, buf, len(buf))
        self.check(n >= 0, "profile_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = (int(calls), float(ms))
        return out

    def close(self):
        if self.h:
            if getattr(self, "_cu_stream", None):
                # (the CU-range stream itself is left to the process: torch may still hold events recorded on its ExternalStream wrapper)
                self.lib.amtgpu_context_synchronize(self.h)
                self.lib.amtgpu_context_set_stream(self.h, None)
                self._cu_stream = None
            self.lib.amtgpu_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class DeviceClip:
    """Frames in HBM: Y (N,H,pitchY), U/V (N,H/2,pitchUV) torch tensors (uint8, or int16/uint16 for >8 bit)."""
    Y: object
    U: object
    V: object
    width: int
    height: int
    bits: int = 8
    fps_num: int = 30000
    fps_den: int = 1001

    @property
    def num_frames(self):
        return int(self.Y.shape[0])

    @property
    def es(self):
        return 1 if self.bits <= 8 else 2

    @property
    def strideY(self):
        return int(self.Y.stride(0)) * self.es

    @property
    def strideUV(self):
        return int(self.U.stride(0)) * self.es

    @property
    def pitchY(self):
        return int(self.Y.stride(1))

    @property
    def pitchUV(self):
        return int(self.U.stride(1))


def weave_fields(ctx: "Context", srcY, srcU, srcV, dst: DeviceClip, top_index=None, bottom_index=None, nv12=False):
    """AMTSource::MakeFrame -> MergeField (AMTSource.hpp:291-366) on decoded pictures in HBM.

    srcY (P,H,pitch), srcU/srcV (P,H/2,pitch) torch tensors (srcU = the interleaved UV plane and srcV = None for NV12);
    dst frame i = even rows of picture top_index[i], odd rows of picture bottom_index[i] (None = i)."""
    es = dst.es
    n = dst.num_frames
    ti = (C.c_int * n)(*[int(v) for v in top_index]) if top_index is not None else None
    bi = (C.c_int * n)(*[int(v) for v in bottom_index]) if bottom_index is not None else None
    ctx.check(ctx.lib.amtgpu_weave_fields_batch(
        ctx.h, _p(srcY), _p(srcU), _p(srcV) if srcV is not None else None, int(srcY.stride(0)) * es, int(srcU.stride(0)) * es,
        int(srcY.stride(1)), int(srcU.stride(1)), int(srcY.shape[0]), ti, bi, 1 if nv12 else 0, dst.bits, dst.width, dst.height,
        _p(dst.Y), _p(dst.U), _p(dst.V), dst.strideY, dst.strideUV, dst.pitchY, dst.pitchUV, n))


class AmtsFile:
    """The stream-index file AMTSource is built from (amts%d.dat; SaveAMTSource / LoadAMTSource, AMTSource.hpp:835-871)."""

    INFO = ("format", "width", "height", "displayWidth", "displayHeight", "sarWidth", "sarHeight", "frameRateNum", "frameRateDenom",
            "colorPrimaries", "transferCharacteristics", "colorSpace", "progressive", "fixedFrameRate", "audioChannels", "sampleRate",
            "decoderMpeg2", "decoderH264", "decoderHevc")

    def __init__(self, path, ctx: "Context" = None):
        self.lib = ctx.lib if ctx else binding.load()
        self.h = self.lib.amtgpu_amts_load(ctx.h if ctx else None, str(path).encode())
        if not self.h:
            raise AmtError(f"cannot read {path}" + (": " + self.lib.amtgpu_last_error(ctx.h).decode(errors="replace") if ctx else ""))
        info = np.zeros(19, np.int32)
        nf, na = C.c_int(), C.c_int()
        self.lib.amtgpu_amts_get_info(self.h, _p(info), C.byref(nf), C.byref(na))
        self.info = dict(zip(self.INFO, map(int, info)))
        self.num_frames, self.num_audio_frames = nf.value, na.value
        cap = 4096
        while True:                                    # (a path is at most 3 UTF-8 bytes per UTF-16 unit: 32 767 units -> < 128 KiB)
            b1, b2 = C.create_string_buffer(cap), C.create_string_buffer(cap)
            if self.lib.amtgpu_amts_get_paths(self.h, b1, cap, b2, cap):
                break
            if cap >= (1 << 20):
                raise AmtError(f"{path}: source / audio path does not fit {cap} bytes")
            cap *= 4
        self.srcpath, self.audiopath = b1.value.decode("utf-8"), b2.value.decode("utf-8")

    def frames(self):
        n = self.num_frames
        out = dict(framePTS=np.zeros(n, np.int64), fileOffset=np.zeros(n, np.int64), keyFrame=np.zeros(n, np.int32),
                   halfDelay=np.zeros(n, np.uint8), cmType=np.zeros(n, np.int32))
        self.lib.amtgpu_amts_get_frames(self.h, _p(out["framePTS"]), _p(out["fileOffset"]), _p(out["keyFrame"]), _p(out["halfDelay"]),
                                        _p(out["cmType"]))
        return out

    def weave_plan(self, picture_pts):
        """(top_index, bottom_index) per frame for decoded pictures with these PTS in output order (AMTSource::OnFrameOutput);
        -1 where the sequence cannot make the frame"""
        pts = np.ascontiguousarray(picture_pts, np.int64)
        top, bot = np.zeros(self.num_frames, np.int32), np.zeros(self.num_frames, np.int32)
        if not self.lib.amtgpu_amts_weave_plan(self.h, _p(pts), len(pts), _p(top), _p(bot)):
            raise AmtError("amtgpu_amts_weave_plan failed")
        return top, bot

    def __del__(self):
        try:
            if self.h:
                self.lib.amtgpu_amts_destroy(self.h)
        except Exception:
            pass


class Logo:
    """logo::LogoData + LogoHeader (AMTLogo.hpp:19-280)."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def load(cls, ctx, path):
        h = ctx.lib.amtgpu_logo_load(ctx.h, str(path).encode())
        ctx.check(h, "LogoData::Load")
        return cls(ctx, h)

    @classmethod
    def from_planes(cls, ctx, planes, w, h, imgw, imgh, imgx, imgy, logUVx=1, logUVy=1):
        planes = np.ascontiguousarray(planes, np.float32)
        hd = ctx.lib.amtgpu_logo_from_planes(ctx.h, w, h, logUVx, logUVy, imgw, imgh, imgx, imgy, _p(planes))
        ctx.check(hd, "logo_from_planes")
        return cls(ctx, hd)

    def save(self, path, name="No Name", service_id=0):
        self.ctx.check(self.ctx.lib.amtgpu_logo_save(self.ctx.h, self.h, str(path).encode(), name.encode(), service_id))

    @property
    def info(self):
        o = np.zeros(8, np.int32)
        self.ctx.lib.amtgpu_logo_get_info(self.h, _p(o))
        return dict(zip(("w", "h", "logUVx", "logUVy", "imgw", "imgh", "imgx", "imgy"), map(int, o)))

    @property
    def planes(self):
        i = self.info
        n = (i["w"] * i["h"] + 2 * (i["w"] >> i["logUVx"]) * (i["h"] >> i["logUVy"])) * 2
        out = np.zeros(n, np.float32)
        self.ctx.lib.amtgpu_logo_get_planes(self.h, _p(out))
        return out

    def mask_tables(self, kind=0, maskratio=0.35):
        i = self.info
        w, h = i["w"], i["h"] if kind == 0 else i["h"] // 2
        mp, cnt, black = C.c_int(), C.c_int(), C.c_float()
        self.ctx.check(self.ctx.lib.amtgpu_logo_mask_tables(self.ctx.h, self.h, kind, maskratio, C.byref(mp), C.byref(cnt), C.byref(black), None, None, None))
        mask = np.zeros(w * h, np.uint8)
        ker = np.zeros(cnt.value * 25, np.float32)
        sc = np.zeros(cnt.value * 64, np.float32)
        self.ctx.check(self.ctx.lib.amtgpu_logo_mask_tables(self.ctx.h, self.h, kind, maskratio, None, None, None, _p(mask), _p(ker), _p(sc)))
        return dict(maskpixels=mp.value, count=cnt.value, blackScore=black.value, mask=mask, kernels=ker, scales=sc)

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.amtgpu_logo_destroy(self.h)
        except Exception:
            pass


class LogoFrame:
    """logo::LogoFrame (LogoScan.hpp:1521-1836)."""

    def __init__(self, ctx: Context, logofiles, maskratio: float):
        self.ctx = ctx
        self.nlogos = len(logofiles)
        if logofiles and isinstance(logofiles[0], Logo):
            arr = (C.c_void_p * self.nlogos)(*[l.h for l in logofiles])
            self._keep = list(logofiles)
            self.h = ctx.lib.amtgpu_logoframe_create_from_logos(ctx.h, arr, self.nlogos, maskratio)
        else:
            arr = (C.c_char_p * self.nlogos)(*[str(p).encode() for p in logofiles])
            self.h = ctx.lib.amtgpu_logoframe_create(ctx.h, arr, self.nlogos, maskratio)
        ctx.check(self.h, "LogoFrame")
        self.num_frames = 0

    def begin(self, width, height, bits, num_frames, fps_num=30000, fps_den=1001):
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_begin(self.h, width, height, bits, num_frames, fps_num, fps_den))
        self.num_frames = num_frames

    def scan_batch(self, Y, bits, first, nframes=None):
        es = 1 if bits <= 8 else 2
        n = int(Y.shape[0]) if nframes is None else nframes
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_scan_batch(self.h, _p(Y), int(Y.stride(0)) * es, int(Y.stride(1)), first, n))

    def scanFrames(self, clip: DeviceClip, batch: int = 4096):
        self.begin(clip.width, clip.height, clip.bits, clip.num_frames, clip.fps_num, clip.fps_den)
        for f0 in range(0, clip.num_frames, batch):
            self.scan_batch(clip.Y[f0:f0 + batch], clip.bits, f0)

    @property
    def evalResults(self):
        out = np.zeros(self.num_frames * self.nlogos * 2, np.float32)
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_get_results(self.h, _p(out)))
        return out.reshape(self.num_frames, self.nlogos, 2)

    def set_results(self, first, evals):
        evals = np.ascontiguousarray(evals, np.float32)
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_set_results(self.h, first, evals.size // (self.nlogos * 2), _p(evals)))

    def selectLogo(self, numCandidates=-1):
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_select_logo(self.h, numCandidates))

    def writeResult(self, outpath, logoIndex=-1):
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_write_result(self.h, str(outpath).encode(), logoIndex))

    def dumpResult(self, basepath):
        """LogoScan.hpp:1632-1643: "<basepath><logo index>", one "%f,%f" line {corr0, corr1} per frame"""
        self.ctx.check(self.ctx.lib.amtgpu_logoframe_dump_result(self.h, str(basepath).encode()))

    def getBestLogo(self):
        return self.ctx.lib.amtgpu_logoframe_best_logo(self.h)

    def getLogoRatio(self):
        return self.ctx.lib.amtgpu_logoframe_logo_ratio(self.h)

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.amtgpu_logoframe_destroy(self.h)
        except Exception:
            pass


def logoframe_decide_host(evals, fps_num=30000, fps_den=1001, numCandidates=-1, logoIndex=-1):
    """LogoFrame::selectLogo + the text of writeResult (LogoScan.hpp:1647-1827) from scan records [frames][logos][2] alone -- host only,
    no device: (bestLogo, logoRatio, text bytes).  What every rank computes after the all-gather of the records."""
    lib = binding.load()
    ev = np.ascontiguousarray(evals, np.float32)
    n, nl = int(ev.shape[0]), int(ev.shape[1])
    best, ratio, tl = C.c_int(-1), C.c_float(0.0), C.c_int(0)
    args = (_p(ev), n, nl, numCandidates, logoIndex, fps_num, fps_den, C.byref(best), C.byref(ratio))
    if lib.amtgpu_logoframe_decide_host(*args, None, 0, C.byref(tl)) != 1:
        raise AmtError("amtgpu_logoframe_decide_host: bad arguments")
    buf = C.create_string_buffer(max(1, tl.value))
    if lib.amtgpu_logoframe_decide_host(*args, buf, tl.value, C.byref(tl)) != 1:
        raise AmtError("amtgpu_logoframe_decide_host failed")
    return best.value, ratio.value, buf.raw[:tl.value]


class AMTAnalyzeLogo:
    """logo::AMTAnalyzeLogo (LogoScan.hpp:1106-1236); GetFrames returns 33 floats per source frame."""

    MODES = {"exact": 0, "linear": 1, "linear_unguarded": 2}

    def __init__(self, ctx: Context, logo, maskratio: float = 0.35, mode: str = "exact"):
        """mode "exact": records bit-identical to the reference's; "linear": all fades from one evaluation of the source and one
        of the background window per mask pixel (scores within `error_bound` of the reference's, decisions guarded by exact
        re-evaluation -- include/amt_gpu.h AMTGPU_ANALYZE_LINEAR_GUARDED)."""
        self.ctx = ctx
        if isinstance(logo, Logo):
            self._keep = logo
            self.h = ctx.lib.amtgpu_analyze_create_from_logo(ctx.h, logo.h, maskratio)
        else:
            self.h = ctx.lib.amtgpu_analyze_create(ctx.h, str(logo).encode(), maskratio)
        ctx.check(self.h, "AMTAnalyzeLogo")
        ctx.check(ctx.lib.amtgpu_analyze_set_mode(self.h, self.MODES[mode]))

    def last_refined(self):
        return self.ctx.lib.amtgpu_analyze_last_refined(self.h)

    def set_fixup_queue(self, entries):
        """pairs a wave of the linear kernel can list for the exact bin check (a tuning knob; results do not depend on it)"""
        self.ctx.check(self.ctx.lib.amtgpu_analyze_set_fixup_queue(self.h, int(entries)))

    def error_bound(self, group=0, bits=8):
        return self.ctx.lib.amtgpu_analyze_error_bound(self.h, group, bits)

    def analyze_device(self, Y, bits, out):
        es = 1 if bits <= 8 else 2
        self.ctx.check(self.ctx.lib.amtgpu_analyze_batch(self.h, _p(Y), int(Y.stride(0)) * es, int(Y.stride(1)), bits, int(Y.shape[0]), _p(out)))

    def analyze(self, clip: DeviceClip):
        out = np.zeros((clip.num_frames, 33), np.float32)
        self.ctx.check(self.ctx.lib.amtgpu_analyze_batch_host(self.h, _p(clip.Y), clip.strideY, clip.pitchY, clip.bits, clip.num_frames, _p(out)))
        return out

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.amtgpu_analyze_destroy(self.h)
        except Exception:
            pass


class AMTEraseLogo:
    """logo::AMTEraseLogo (LogoScan.hpp:1238-1519), mode 0."""

    def __init__(self, ctx: Context, logo, logof="", mode: int = 0, maxfade: int = 16):
        self.ctx = ctx
        if isinstance(logo, Logo):
            self._keep = logo
            self.h = ctx.lib.amtgpu_erase_create_from_logo(ctx.h, logo.h, (logof or "").encode(), mode, maxfade)
        else:
            self.h = ctx.lib.amtgpu_erase_create(ctx.h, str(logo).encode(), str(logof or "").encode(), mode, maxfade)
        ctx.check(self.h, "AMTEraseLogo")

    def calc_fades(self, analysis, num_frames, first=0, nframes=None):
        analysis = np.ascontiguousarray(analysis, np.float32)
        n = num_frames - first if nframes is None else nframes
        out = np.zeros((n, 2), np.float32)
        self.ctx.check(self.ctx.lib.amtgpu_erase_calc_fades(self.h, _p(analysis), num_frames, first, n, _p(out)))
        return out

    def calc_fades_device(self, d_analysis, num_frames, first=0, nframes=None, analysis_first=0, out=None):
        """CalcFade / CalcFade2 on the device: d_analysis = torch float32 [count, 33] records of source frames
        [analysis_first, analysis_first + count) still in HBM; returns a torch float32 [nframes, 2] tensor (async)."""
        import torch
        n = num_frames - first if nframes is None else nframes
        if out is None:
            out = torch.empty((n, 2), dtype=torch.float32, device=d_analysis.device)
        self.ctx.check(self.ctx.lib.amtgpu_erase_calc_fades_device(self.h, _p(d_analysis), analysis_first, int(d_analysis.shape[0]), num_frames,
                                                                   first, n, _p(out)))
        return out

    def erase_device_fades(self, clip: DeviceClip, d_fades, dst: "DeviceClip | None" = None):
        """Delogo with fades that are already on the device (calc_fades_device's output).  async
        dst: a batch of the same geometry that already holds a copy of clip's frames -- the writable copy AMTEraseLogo::GetFrameT takes
        (env->MakeWritable, LogoScan.hpp:1346-1347): Delogo reads clip and writes dst's rectangle, clip stays intact.  None: in place."""
        if dst is None:
            self.ctx.check(self.ctx.lib.amtgpu_erase_batch_dfades(self.h, _p(clip.Y), _p(clip.U), _p(clip.V), clip.strideY, clip.strideUV,
                                                                  clip.pitchY, clip.pitchUV, clip.bits, clip.num_frames, _p(d_fades)))
            return
        same = ("strideY", "strideUV", "pitchY", "pitchUV", "bits", "num_frames")
        if any(getattr(clip, k) != getattr(dst, k) for k in same):
            raise ValueError("erase_device_fades: source and destination batches differ in geometry")
        self.ctx.check(self.ctx.lib.amtgpu_erase_batch_dfades_to(self.h, _p(clip.Y), _p(clip.U), _p(clip.V), _p(dst.Y), _p(dst.U), _p(dst.V),
                                                                 clip.strideY, clip.strideUV, clip.pitchY, clip.pitchUV, clip.bits,
                                                                 clip.num_frames, _p(d_fades)))

    def erase(self, clip: DeviceClip, fades):
        fades = np.ascontiguousarray(fades, np.float32)
        self.ctx.check(self.ctx.lib.amtgpu_erase_batch(self.h, _p(clip.Y), _p(clip.U), _p(clip.V), clip.strideY, clip.strideUV,
                                                       clip.pitchY, clip.pitchUV, clip.bits, clip.num_frames, _p(fades)))

    @property
    def rect(self):
        """(imgx, imgy, w, h, fade0_is_identity): the rectangle Delogo rewrites"""
        out = (C.c_int * 5)()
        self.ctx.check(self.ctx.lib.amtgpu_erase_get_rect(self.h, out), "erase_get_rect")
        return tuple(out)

    def erase_rect(self, Y, U, V, bits, fades):
        """Delogo on device planes that hold only the logo rectangle: Y [n, h, w], U / V [n, h/2, w/2] (any row pitch)"""
        fades = np.ascontiguousarray(fades, np.float32)
        es = 1 if bits <= 8 else 2
        self.ctx.check(self.ctx.lib.amtgpu_erase_rect_batch(self.h, _p(Y), _p(U), _p(V), int(Y.stride(0)) * es, int(U.stride(0)) * es,
                                                            int(Y.stride(1)), int(U.stride(1)), bits, int(Y.shape[0]), _p(fades)))

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.amtgpu_erase_destroy(self.h)
        except Exception:
            pass


class LogoScan:
    """logo::LogoScan (LogoScan.hpp:398-660) with exact integer accumulators."""

    def __init__(self, ctx: Context, w, h, thy, logUVx=1, logUVy=1):
        self.ctx, self.w, self.hh = ctx, w, h
        self.h = ctx.lib.amtgpu_logoscan_create(ctx.h, w, h, logUVx, logUVy, thy)
        ctx.check(self.h, "LogoScan")

    def add_batch(self, clip: DeviceClip, imgx, imgy, max_valid=1 << 30, use_mask=None):
        valid = np.zeros(clip.num_frames, np.uint8)
        n = C.c_int()
        um = None if use_mask is None else np.ascontiguousarray(use_mask, np.uint8)
        self.ctx.check(self.ctx.lib.amtgpu_logoscan_add_batch(self.h, _p(clip.Y), _p(clip.U), _p(clip.V), clip.strideY, clip.strideUV,
                                                              clip.pitchY, clip.pitchUV, clip.bits, imgx, imgy, clip.num_frames,
                                                              max_valid, _p(um), _p(valid), C.byref(n)))
        return valid, n.value

    @property
    def nframes(self):
        return self.ctx.lib.amtgpu_logoscan_nframes(self.h)

    def sums(self):
        npx = self.w * self.hh + 2 * (self.w // 2) * (self.hh // 2)
        s = np.zeros(npx * 3, np.int64)
        p = np.zeros(6, np.int64)
        self.ctx.check(self.ctx.lib.amtgpu_logoscan_get_sums(self.h, _p(s), _p(p)))
        return s, p

    def set_sums(self, s, p, nframes):
        s = np.ascontiguousarray(s, np.int64)
        p = np.ascontiguousarray(p, np.int64)
        self.ctx.check(self.ctx.lib.amtgpu_logoscan_set_sums(self.h, _p(s), _p(p), nframes))

    def get_logo(self, maxv, clean, imgw, imgh, imgx, imgy):
        h = self.ctx.lib.amtgpu_logoscan_get_logo(self.h, maxv, 1 if clean else 0, imgw, imgh, imgx, imgy)
        self.ctx.check(h, "LogoScan::GetLogo")
        return Logo(self.ctx, h)

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.amtgpu_logoscan_destroy(self.h)
        except Exception:
            pass


def ScanLogo(ctx: Context, clip: DeviceClip, serviceid, dstpath, imgx, imgy, w, h, thy, numMaxFrames, cb=None):
    """The exported ScanLogo (LogoScan.hpp:1083-1098) over a device clip; returns True/False like it."""
    cbf = binding.CB(cb) if cb else binding.CB(lambda p, a, b, c: 1)
    ok = ctx.lib.amtgpu_scanlogo(ctx.h, _p(clip.Y), _p(clip.U), _p(clip.V), clip.strideY, clip.strideUV, clip.pitchY, clip.pitchUV,
                                 clip.width, clip.height, clip.num_frames, serviceid, str(dstpath).encode(), imgx, imgy, w, h, thy,
                                 numMaxFrames, cbf)
    return bool(ok)


def ScanLogoFile(ctx: Context, srcpath, serviceid, workfile, dstpath, imgx, imgy, w, h, thy, numMaxFrames, cb=None):
    """ScanLogo with the reference's own argument list (LogoScan.hpp:1083-1098) over a raw 'AMTR' clip file; True/False like it."""
    cbf = binding.CB(cb) if cb else binding.CB(lambda p, a, b, c: 1)
    return bool(ctx.lib.amtgpu_scanlogo_file(ctx.h, str(srcpath).encode(), serviceid, str(workfile).encode(), str(dstpath).encode(), imgx, imgy,
                                             w, h, thy, numMaxFrames, cbf))


class FrameStats:
    """Self-specified whole-frame field-difference / combing metrics."""

    def __init__(self, ctx: Context, width, height, bits=8):
        self.ctx, self.width, self.height, self.bits = ctx, width, height, bits
        self.h = ctx.lib.amtgpu_framestats_create(ctx.h, width, height, bits)
        ctx.check(self.h, "FrameStats")

    def run_device(self, Y, out, prevY=None):
        es = 1 if self.bits <= 8 else 2
        self.ctx.check(self.ctx.lib.amtgpu_framestats_batch(self.h, _p(Y), int(Y.stride(0)) * es, int(Y.stride(1)), _p(prevY), int(Y.shape[0]), _p(out)))

    def run(self, clip: DeviceClip):
        import torch
        out = torch.zeros((clip.num_frames, 8), dtype=torch.int64, device=clip.Y.device)
        self.run_device(clip.Y, out)
        self.ctx.synchronize()
        return out.cpu().numpy().astype(np.uint64)

    def scene_changes(self, metrics):
        m = np.ascontiguousarray(metrics, np.uint64)
        n = m.shape[0]
        out = np.zeros(max(1, n), np.int32)
        k = C.c_int()
        self.ctx.lib.amtgpu_cm_scene_changes(_p(m), n, self.width, self.height, _p(out), n, C.byref(k))
        return out[:k.value].copy()

    def cadence(self, metrics):
        m = np.ascontiguousarray(metrics, np.uint64)
        n = m.shape[0]
        cad = np.zeros(n, np.uint8)
        ph = np.zeros(n, np.uint8)
        self.ctx.lib.amtgpu_kfm_cadence(_p(m), n, self.width, self.height, _p(cad), _p(ph))
        return cad, ph

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.amtgpu_framestats_destroy(self.h)
        except Exception:
            pass
