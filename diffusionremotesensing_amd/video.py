"""`video_maker` stand-in for the reference's mp4 writer (utils.py:384-430, cv2 + PIL drawing).  Encoding a video is
outside the denoising hot path and cv2 is not part of this build: when cv2 is importable the frames are written as the
reference does (mp4v, first image of every frame, clamped to [0, 1] * 255, RGB->BGR, without the frame-number overlay);
otherwise the uint8 frames are saved next to the requested path as `<path>.frames.pt` so that `generate_video=True`
keeps working."""
import torch


def video_maker(frames, video_path="output.mp4", fps=50):
    if not frames:
        return
    if float(frames[0].max()) < 100:  # reference :395-396
        frames = [torch.clamp(f[0], 0, 1) * 255 for f in frames]
    frames = [f.to(torch.uint8).permute(1, 2, 0).detach().cpu() for f in frames]
    print("Creating video... with frames:", len(frames))
    try:
        import cv2
    except ImportError:
        torch.save(torch.stack(frames), video_path + ".frames.pt")
        print(f"cv2 is not available: {len(frames)} uint8 frames saved to {video_path}.frames.pt")
        return
    height, width = frames[0].shape[:2]
    video = cv2.VideoWriter(video_path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (width, height))
    for f in frames:
        video.write(cv2.cvtColor(f.numpy(), cv2.COLOR_RGB2BGR))
    video.release()
