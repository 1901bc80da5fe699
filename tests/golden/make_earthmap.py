"""Decode the reference's earthmap.jpg (a data file of the reference, used by
src/texture.rs:91 and src/main.rs:155,286) into a raw binary PPM fixture.

Run once in the build container (needs /root/reference and Pillow); the GPU
box only ever reads the committed tests/golden/earthmap.ppm.  The reference
decodes with the pure-Rust `jpeg-decoder`; Pillow uses libjpeg-turbo, whose
IDCT/upsampling may differ by an LSB or two on some texels (parity unpinned at
this boundary; far below render noise).
"""
import hashlib
import sys
from pathlib import Path

from PIL import Image

src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/raytracer-rs/earthmap.jpg")
dst = Path(__file__).resolve().parent / "earthmap.ppm"
raw = src.read_bytes()
print("jpeg sha256", hashlib.sha256(raw).hexdigest())
im = Image.open(src).convert("RGB")
w, h = im.size
data = im.tobytes()
dst.write_bytes(b"P6\n%d %d\n255\n" % (w, h) + data)
print("wrote", dst, w, h, "rgb sha256", hashlib.sha256(data).hexdigest())
