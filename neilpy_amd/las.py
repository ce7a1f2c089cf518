"""LAS point-cloud input for the SMRF path (reference: neilpy/neilpy.py:903-1087).

``read_las(filename)`` is the drop-in: ``(header dict, pandas DataFrame)`` with the reference's
column names, dtypes and bit-field decoding for point data record formats 0-10 of LAS 1.0-1.4
(LAZ is rejected, as in the reference).  ``read_las_xyz(filename)`` is the GPU entry that feeds
``create_dem`` / ``smrf``: the header is parsed on the host, the raw point records go to HBM once
and a HIP kernel (``smrf_las_decode_xyz_f64``) turns the packed int32 coordinates into float64
``x, y, z`` tensors (``value * scale + offset``, product and sum rounded separately like the
reference's pandas arithmetic).

Record layouts follow the ASPRS LAS 1.4 R15 specification (tables 7-17); only the fields the
reference exposes are named.
"""
import struct

import numpy as np

__all__ = ["read_las", "read_las_xyz", "write_las", "parse_header", "record_dtype"]

# minimum record size per point data record format (LAS 1.4 R15, section 2.6)
RECORD_SIZE = {0: 20, 1: 28, 2: 26, 3: 34, 4: 57, 5: 63, 6: 30, 7: 36, 8: 38, 9: 59, 10: 67}

_CORE_LEGACY = [('x', '<i4'), ('y', '<i4'), ('z', '<i4'), ('intensity', '<u2'), ('return_byte', 'u1'),
                ('class', 'u1'), ('scan_angle', 'u1'), ('user_data', 'u1'), ('point_source_id', '<u2')]
_CORE_14 = [('x', '<i4'), ('y', '<i4'), ('z', '<i4'), ('intensity', '<u2'), ('return_byte', 'u1'),
            ('mixed_byte', 'u1'), ('class', 'u1'), ('user_data', 'u1'), ('scan_angle', '<u2'),
            ('point_source_id', '<u2'), ('gpstime', '<f8')]
_GPS = [('gpstime', '<f8')]
_RGB = [('red', '<u2'), ('green', '<u2'), ('blue', '<u2')]
_NIR = [('near_infrared', '<u2')]
_WAVE = [('wave_packet_descriptor_index', 'u1'), ('byte_offset', '<u8'), ('wave_packet_size', '<u4'),
         ('return_point_waveform_location', '<f4'), ('xt', '<f4'), ('yt', '<f4'), ('zt', '<f4')]
_FIELDS = {0: _CORE_LEGACY, 1: _CORE_LEGACY + _GPS, 2: _CORE_LEGACY + _RGB, 3: _CORE_LEGACY + _GPS + _RGB,
           4: _CORE_LEGACY + _GPS + _WAVE, 5: _CORE_LEGACY + _GPS + _RGB + _WAVE,
           6: _CORE_14, 7: _CORE_14 + _RGB, 8: _CORE_14 + _RGB + _NIR, 9: _CORE_14 + _WAVE,
           10: _CORE_14 + _RGB + _NIR + _WAVE}


def record_dtype(fmt, record_length=None):
    """Packed little-endian dtype of one point record; extra bytes beyond the format's fields are
    kept as padding so that files with user-defined extra bytes still parse."""
    fields = list(_FIELDS[fmt])
    size = RECORD_SIZE[fmt]
    assert np.dtype(fields).itemsize == size
    if record_length is not None and record_length > size:
        fields.append(('extra_bytes', 'V%d' % (record_length - size)))
    return np.dtype(fields)


def parse_header(data, total_size=None):
    """Public header block -> dict with the reference's keys (neilpy.py:926-973).  ``data`` may be the
    file's first bytes only when ``total_size`` gives the file length."""
    u = struct.unpack_from
    h = {}
    h['file_signature'] = u('<4s', data, 0)[0].decode('utf-8')
    h['file_source_id'] = u('<H', data, 4)[0]
    h['global_encoding'] = u('<H', data, 6)[0]
    h['project_id'] = [u('<L', data, 8)[0], u('<H', data, 12)[0], u('<H', data, 14)[0]]
    h['version_major'] = u('<B', data, 24)[0]
    h['version_minor'] = u('<B', data, 25)[0]
    h['version'] = h['version_major'] + h['version_minor'] / 10
    h['system_id'] = u('32s', data, 26)[0].decode('utf-8').rstrip('\x00')
    h['generating_software'] = u('32s', data, 58)[0].decode('utf-8').rstrip('\x00')
    h['file_creation_day'] = u('<H', data, 90)[0]
    h['file_creation_year'] = u('<H', data, 92)[0]
    h['header_size'] = u('<H', data, 94)[0]
    h['point_data_offset'] = u('<L', data, 96)[0]
    h['num_variable_records'] = u('<L', data, 100)[0]
    fmt = u('<B', data, 104)[0]
    if 128 <= fmt <= 133:
        raise ValueError('LAZ not yet supported.')
    h['point_data_format_id'] = fmt
    if fmt not in RECORD_SIZE:
        raise ValueError('Point Data Record Format', fmt, 'not yet supported.')
    h['point_data_record_length'] = u('<H', data, 105)[0]
    h['num_point_records'] = u('<L', data, 107)[0]
    h['num_points_by_return'] = u('<5L', data, 111)
    h['scale'] = u('<3d', data, 131)
    h['offset'] = u('<3d', data, 155)
    h['minmax'] = u('<6d', data, 179)                 # xmax, xmin, ymax, ymin, zmax, zmin
    end = len(data) if total_size is None else total_size
    if h['version'] == 1.3:
        h['begin_wave_form'] = u('<q', data, 227)[0]
        if h['begin_wave_form'] != 0:
            end = h['begin_wave_form']
    return h, end


def _bit(v, i):
    return (v & (1 << i)) != 0


def _load(filename):
    with open(filename, mode='rb') as f:
        data = f.read()
    header, end = parse_header(data)
    fmt = header['point_data_format_id']
    reclen = max(header['point_data_record_length'], RECORD_SIZE[fmt])
    raw = memoryview(data)[header['point_data_offset']:end]
    npts = len(raw) // reclen
    return header, raw[:npts * reclen], reclen, npts


def read_las(filename):
    """Reads a LAS (not LAZ) file into ``(header, DataFrame)`` - the reference's signature and columns."""
    import pandas as pd
    header, raw, reclen, npts = _load(filename)
    fmt = header['point_data_format_id']
    rec = np.frombuffer(raw, dtype=record_dtype(fmt, reclen), count=npts)
    names = [n for n in rec.dtype.names if n != 'extra_bytes']
    data = pd.DataFrame({n: rec[n] for n in names})
    for i, axis in enumerate('xyz'):
        data[axis] = data[axis] * header['scale'][i] + header['offset'][i]
    rb = data['return_byte']
    u8 = np.uint8
    if fmt < 6:
        data['return_number'] = 4 * _bit(rb, 2).astype(u8) + 2 * _bit(rb, 1).astype(u8) + _bit(rb, 0).astype(u8)
        data['return_max'] = 4 * _bit(rb, 5).astype(u8) + 2 * _bit(rb, 4).astype(u8) + _bit(rb, 3).astype(u8)
        data['scan_direction'] = _bit(rb, 6)
        data['edge_of_flight_line'] = _bit(rb, 7)
        del data['return_byte']
    else:
        data['return_number'] = (8 * _bit(rb, 3).astype(u8) + 4 * _bit(rb, 2).astype(u8) + 2 * _bit(rb, 1).astype(u8)
                                 + _bit(rb, 0).astype(u8))
        data['return_max'] = (8 * _bit(rb, 7).astype(u8) + 4 * _bit(rb, 6).astype(u8) + 2 * _bit(rb, 5).astype(u8)
                              + _bit(rb, 4).astype(u8))
        del data['return_byte']
        mb = data['mixed_byte']
        data['classification_bit_synthetic'] = _bit(mb, 0)
        data['classification_bit_keypoint'] = _bit(mb, 1)
        data['classification_bit_withheld'] = _bit(mb, 2)
        data['classification_bit_overlap'] = _bit(mb, 3)
        data['scanner_channel'] = 2 * _bit(mb, 5).astype(u8) + 1 * _bit(mb, 4).astype(u8)
        data['scan_direction'] = _bit(mb, 6)
        data['edge_of_flight_line'] = _bit(mb, 7)
        del data['mixed_byte']
    return header, data


def read_las_xyz(filename):
    """``(header, x, y, z)`` with x, y, z float64 CUDA tensors decoded on the GPU from the raw records.

    The point records are streamed file -> pinned staging buffer -> device in 16 MB chunks (the read of
    chunk k+1 overlaps the copy of chunk k), then one kernel applies scale and offset (neilpy.py:1055-1057)."""
    import ctypes as C
    import os
    import torch
    from . import _lib
    _lib.require_gpu()
    size = os.path.getsize(filename)
    dev = torch.device("cuda", torch.cuda.current_device())
    with open(filename, mode='rb') as f:
        header, end = parse_header(f.read(min(size, 1024)), total_size=size)
        fmt = header['point_data_format_id']
        reclen = max(header['point_data_record_length'], RECORD_SIZE[fmt])
        start = header['point_data_offset']
        npts = max(0, min(end, size) - start) // reclen
        nbytes = npts * reclen
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:
            from ._xfer import STAGE_BYTES, staging
            stage, events = staging()
            f.seek(start)
            pos = k = 0
            while pos < nbytes:
                n = min(STAGE_BYTES, nbytes - pos)
                events[k % 2].synchronize()                      # the copy that last used this buffer is done
                got = f.readinto(memoryview(stage[k % 2].numpy())[:n])
                if got != n:
                    raise ValueError("LAS file truncated: expected %d more bytes of point records" % (n - got))
                buf[pos:pos + n].copy_(stage[k % 2][:n], non_blocking=True)
                events[k % 2].record()
                pos += n
                k += 1
    out = [torch.empty(npts, dtype=torch.float64, device=dev) for _ in range(3)]
    so = (C.c_double * 6)(*(list(header['scale']) + list(header['offset'])))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().smrf_las_decode_xyz_f64(C.c_void_p(buf.data_ptr()), npts, reclen, so,
                                                   *(C.c_void_p(t.data_ptr()) for t in out), st))
    return (header,) + tuple(out)


def write_las(filename, x, y, z, fmt=1, scale=(0.01, 0.01, 0.01), offset=None, version=(1, 2), extra_bytes=0,
              fields=None, system_id="neilpy_amd", software="neilpy_amd.las.write_las"):
    """Minimal LAS writer (tests, stand-in inputs): one public header block, no VLRs.
    ``fields`` may give further record fields by name (e.g. ``{'intensity': arr, 'return_byte': arr}``)."""
    x, y, z = (np.asarray(v, dtype=np.float64) for v in (x, y, z))
    n = x.size
    if offset is None:
        offset = (float(np.floor(x.min())) if n else 0.0, float(np.floor(y.min())) if n else 0.0,
                  float(np.floor(z.min())) if n else 0.0)
    reclen = RECORD_SIZE[fmt] + extra_bytes
    rec = np.zeros(n, dtype=record_dtype(fmt, reclen))
    for i, (axis, v) in enumerate(zip('xyz', (x, y, z))):
        rec[axis] = np.round((v - offset[i]) / scale[i]).astype(np.int64)
    for k, v in (fields or {}).items():
        rec[k] = v
    header_size = 227 if version < (1, 3) else (235 if version == (1, 3) else 375)
    hdr = bytearray(header_size)
    p = struct.pack_into
    p('<4s', hdr, 0, b'LASF')
    p('<BB', hdr, 24, version[0], version[1])
    p('32s', hdr, 26, system_id.encode())
    p('32s', hdr, 58, software.encode())
    p('<HH', hdr, 90, 1, 2024)
    p('<H', hdr, 94, header_size)
    p('<L', hdr, 96, header_size)
    p('<L', hdr, 100, 0)
    p('<B', hdr, 104, fmt)
    p('<H', hdr, 105, reclen)
    p('<L', hdr, 107, n if n < 2 ** 32 else 0)
    p('<5L', hdr, 111, n if n < 2 ** 32 else 0, 0, 0, 0, 0)
    p('<3d', hdr, 131, *scale)
    p('<3d', hdr, 155, *offset)
    if n:
        dx, dy, dz = (rec[a] * scale[i] + offset[i] for i, a in enumerate('xyz'))
        p('<6d', hdr, 179, dx.max(), dx.min(), dy.max(), dy.min(), dz.max(), dz.min())
    with open(filename, 'wb') as f:
        f.write(bytes(hdr))
        f.write(rec.tobytes())
