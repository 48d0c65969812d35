"""Test-side writer of an `amts%d.dat` stream-index file in the layout SaveAMTSource produces on the reference's platform
(AMTSource.hpp:835-852: File::writeArray = int64 count + raw elements, writeValue = raw struct; MSVC x64 struct layouts,
2-byte wchar_t), and a plain-Python restatement of AMTSource::OnFrameOutput's picture -> frame matching (:482-566)."""
import struct


def write_amts(path, srcpath, audiopath, vfmt, afmt, frames, audio_frames, decoder=(0, 0, 0)):
    """vfmt: (format,width,height,displayWidth,displayHeight,sarWidth,sarHeight,frameRateNum,frameRateDenom,
    colorPrimaries,transferCharacteristics,colorSpace,progressive,fixedFrameRate); afmt: (channels, sampleRate);
    frames: dicts halfDelay, frameIndex, pts, frameDuration, framePTS, fileOffset, keyFrame, cmType;
    audio_frames: (frameIndex, waveOffset, waveLength)"""
    with open(path, "wb") as f:
        for s in (srcpath, audiopath):
            u = s.encode("utf-16-le")
            f.write(struct.pack("<q", len(u) // 2) + u)
        f.write(struct.pack("<9i3B2?3x", *vfmt))                       # VideoFormat: 44 bytes
        f.write(struct.pack("<2i", *afmt))                             # AudioFormat: 8 bytes
        f.write(struct.pack("<q", len(frames)))
        for fr in frames:                                              # FilterSourceFrame: 48 bytes
            f.write(struct.pack("<?3xiddqqii", fr["halfDelay"], fr["frameIndex"], fr["pts"], fr["frameDuration"], fr["framePTS"],
                                fr["fileOffset"], fr["keyFrame"], fr["cmType"]))
        f.write(struct.pack("<q", len(audio_frames)))
        for a in audio_frames:                                         # FilterAudioFrame: 24 bytes
            f.write(struct.pack("<i4xqi4x", *a))
        f.write(struct.pack("<3i", *decoder))                          # DecoderSetting: 12 bytes


def reference_plan(frame_pts, half_delay, picture_pts):
    """AMTSource::OnFrameOutput (AMTSource.hpp:482-566): returns (top, bottom) picture index per frame, -1 where none is made"""
    import bisect
    nf = len(frame_pts)
    top, bot = [-1] * nf, [-1] * nf
    prev = None
    for k, p in enumerate(picture_pts):
        pts = p & ((1 << 33) - 1)
        it = bisect.bisect_left(frame_pts, pts)
        if it == 0 and pts < frame_pts[0]:
            pts += 1 << 33
            it = bisect.bisect_left(frame_pts, pts)
        if it == nf or frame_pts[it] != pts:
            prev = None
            continue
        if half_delay[it]:
            if top[it] < 0 and prev is not None:
                top[it], bot[it] = prev, k
            if it + 1 < nf and frame_pts[it + 1] == frame_pts[it] and top[it + 1] < 0:
                top[it + 1], bot[it + 1] = k, k
        elif top[it] < 0:
            top[it], bot[it] = k, k
        prev = k
    return top, bot
