"""On-disk bitstream container of the reference harness (utils/IOutils.py:30-91), byte for byte.

Per modality file: BE u32 H, W (original size) | BE u32 shape[0], shape[1], n_lists | per list: BE u32 count |
per string: BE u32 length + raw bytes.   bpp = filesize * 8 / (H * W)   (testing/tester_united.py:153-165)
"""
import struct
from pathlib import Path


def write_uints(fd, values, fmt=">{:d}I"):
    fd.write(struct.pack(fmt.format(len(values)), *values))
    return len(values) * 4


def read_uints(fd, n, fmt=">{:d}I"):
    return struct.unpack(fmt.format(n), fd.read(n * 4))


def write_bytes(fd, values):
    if len(values) == 0:
        return 0
    fd.write(bytes(values))
    return len(values)


def read_bytes(fd, n):
    return fd.read(n)


def write_body(fd, shape, out_strings):
    cnt = write_uints(fd, (int(shape[0]), int(shape[1]), len(out_strings)))
    for lst in out_strings:
        cnt += write_uints(fd, (len(lst),))
        for s in lst:
            cnt += write_uints(fd, (len(s),))
            cnt += write_bytes(fd, s)
    return cnt


def read_body(fd):
    shape = read_uints(fd, 2)
    n_lists = read_uints(fd, 1)[0]
    lstrings = []
    for _ in range(n_lists):
        num = read_uints(fd, 1)[0]
        lstrings.append([read_bytes(fd, read_uints(fd, 1)[0]) for _ in range(num)])
    return lstrings, shape


def filesize(filepath) -> int:
    p = Path(filepath)
    if not p.is_file():
        raise ValueError(f'Invalid file "{filepath}".')
    return p.stat().st_size
