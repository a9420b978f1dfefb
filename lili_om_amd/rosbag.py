"""Pure-Python ROS bag (format 2.0) reader and the three message types the reference's nodes subscribe to
(SURVEY §8 f-4): sensor_msgs/PointCloud2 (`/livox_ros_points`, lidar_topic), sensor_msgs/Imu and
livox_ros_driver/CustomMsg (`/livox/lidar`, L/src/FormatConvert.cpp:49).  No ROS installation is needed: this is
the replay side of the drop-in boundary — decoded point buffers go to the C ABI as they are
(`lili_cloud{data, n, point_step, offset(intensity), LILI_MEM_HOST}` / `lili_livox_custom_to_cloud`).

Bag format 2.0 (http://wiki.ros.org/Bags/Format/2.0): "#ROSBAG V2.0\\n", then records
    <uint32 header_len> <header: (uint32 field_len, "name=value")*> <uint32 data_len> <data>
with op = 0x03 bag header, 0x05 chunk (compression none / bz2 / lz4), 0x07 connection, 0x02 message data,
0x04 index data, 0x06 chunk info.  Messages are yielded in file order (chunk order, then position in the chunk),
which is the order `rosbag play` delivers them for a single-writer bag.  All integers little-endian.

The small writer at the bottom exists for the round-trip tests (no bag files can be shipped offline); it writes
uncompressed or bz2 chunks with connection records and message data, without the optional index records.
"""
import bz2
import struct

import numpy as np

MAGIC = b"#ROSBAG V2.0\n"
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07

CUSTOM_POINT = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
                         ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1")])   # 19 bytes, packed


class BagError(ValueError):
    pass


def _parse_header(buf):
    fields, pos = {}, 0
    while pos < len(buf):
        if pos + 4 > len(buf):
            raise BagError("truncated record header")
        (n,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        kv = buf[pos:pos + n]
        if len(kv) != n or b"=" not in kv:
            raise BagError("malformed header field")
        k, v = kv.split(b"=", 1)
        fields[k.decode()] = v
        pos += n
    return fields


def _records(buf, pos=0, end=None):
    end = len(buf) if end is None else end
    while pos < end:
        if pos + 4 > end:
            raise BagError("truncated record")
        (hl,) = struct.unpack_from("<I", buf, pos)
        hdr = _parse_header(bytes(buf[pos + 4:pos + 4 + hl]))
        pos += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, pos)
        data = buf[pos + 4:pos + 4 + dl]
        if len(data) != dl:
            raise BagError("truncated record data")
        pos += 4 + dl
        yield hdr, data


def _lz4_block(src, out_size):
    """LZ4 block decompression (sequences of literals + matches), enough for ROS' lz4 chunks' inner blocks."""
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        if off == 0 or off > len(out):
            raise BagError("corrupt lz4 block")
        start = len(out) - off
        for k in range(ml):                      # overlapping copies are legal
            out.append(out[start + k])
    if out_size is not None and len(out) != out_size:
        raise BagError("lz4 size mismatch")
    return bytes(out)


def _roslz4(data, size):
    """ROS' lz4 chunk = an LZ4 *frame* (magic 0x184D2204) as written by roslz4; decode its blocks."""
    if data[:4] != b"\x04\x22\x4d\x18":
        raise BagError("not an lz4 frame")
    flg = data[4]
    pos = 6 + (8 if flg & 0x08 else 0) + 1         # FLG, BD, [content size], HC
    out = bytearray()
    while True:
        (bs,) = struct.unpack_from("<I", data, pos); pos += 4
        if bs == 0:
            break
        raw = bool(bs & 0x80000000)
        bs &= 0x7FFFFFFF
        blk = data[pos:pos + bs]; pos += bs
        out += blk if raw else _lz4_block(blk, None)
        if flg & 0x10:
            pos += 4                              # block checksum
    if len(out) != size:
        raise BagError("lz4 chunk size mismatch")
    return bytes(out)


class Bag:
    """for topic, msgtype, t, raw in Bag(path).messages(topics=[...]): ...   (t = receive time in seconds, float)"""

    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        if not self.buf.startswith(MAGIC):
            raise BagError("not a ROS bag format 2.0 file")
        self.connections = {}

    def _conn(self, hdr, data):
        cid = struct.unpack("<I", hdr["conn"])[0]
        info = _parse_header(bytes(data))
        self.connections[cid] = dict(topic=hdr["topic"].decode(), type=info.get("type", b"").decode(), md5=info.get("md5sum", b"").decode())

    def messages(self, topics=None):
        want = None if topics is None else set(topics)
        for hdr, data in _records(self.buf, len(MAGIC)):
            op = hdr["op"][0]
            if op == OP_CONNECTION:
                self._conn(hdr, data)
            elif op == OP_CHUNK:
                comp = hdr["compression"].decode()
                size = struct.unpack("<I", hdr["size"])[0]
                if comp == "none":
                    chunk = data
                elif comp == "bz2":
                    chunk = bz2.decompress(bytes(data))
                elif comp == "lz4":
                    chunk = _roslz4(bytes(data), size)
                else:
                    raise BagError(f"unknown chunk compression {comp}")
                if len(chunk) != size:
                    raise BagError("chunk size mismatch")
                for h2, d2 in _records(chunk):
                    op2 = h2["op"][0]
                    if op2 == OP_CONNECTION:
                        self._conn(h2, d2)
                    elif op2 == OP_MSG:
                        cid = struct.unpack("<I", h2["conn"])[0]
                        c = self.connections.get(cid)
                        if c is None:
                            raise BagError("message before its connection record")
                        if want is not None and c["topic"] not in want:
                            continue
                        sec, nsec = struct.unpack("<II", h2["time"])
                        yield c["topic"], c["type"], sec + nsec * 1e-9, bytes(d2)
            # bag header, index data, chunk info: not needed for sequential replay


# ---------------------------------------------------------------------------------------------------
# message (de)serialisation — ROS1 wire format: little endian, strings / arrays length-prefixed with uint32
# ---------------------------------------------------------------------------------------------------
class _R:
    def __init__(self, b):
        self.b, self.p = b, 0

    def u(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def s(self):
        n = self.u("I")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v.decode(errors="replace")

    def raw(self, n):
        v = self.b[self.p:self.p + n]
        if len(v) != n:
            raise BagError("truncated message")
        self.p += n
        return v


def _header(r):
    seq = r.u("I")
    sec, nsec = r.u("II")
    return dict(seq=seq, stamp=sec + nsec * 1e-9, stamp_sec=sec, stamp_nsec=nsec, frame_id=r.s())


def decode_pointcloud2(raw):
    """sensor_msgs/PointCloud2 -> dict(header, height, width, fields {name: (offset, datatype, count)}, point_step,
    row_step, is_bigendian, is_dense, data = uint8 array of width*height records of point_step bytes).
    The PCL layouts of the reference: PointXYZINormal x0 y4 z8 | normal 16 20 24 | intensity 32 curvature 36, step 48;
    PointXYZI x0 y4 z8 intensity 16, step 32."""
    r = _R(raw)
    h = _header(r)
    height, width = r.u("II")
    nf = r.u("I")
    fields = {}
    for _ in range(nf):
        name = r.s()
        off, dt, cnt = r.u("IBI")
        fields[name] = (off, dt, cnt)
    big = bool(r.u("B"))
    point_step, row_step = r.u("II")
    n = r.u("I")
    data = np.frombuffer(r.raw(n), np.uint8)
    dense = bool(r.u("B"))
    if big:
        raise BagError("big-endian PointCloud2 is not supported")
    return dict(header=h, height=height, width=width, fields=fields, point_step=point_step, row_step=row_step,
                is_bigendian=big, is_dense=dense, data=data)


def pointcloud2_xyz_aux(msg, aux="intensity", extra=None):
    """View the cloud as float32 columns x, y, z, aux[, extra] (copy, (n, 4|5)); FLOAT32 fields only (datatype 7)."""
    n = msg["width"] * msg["height"]
    step = msg["point_step"]
    rec = msg["data"][:n * step].reshape(n, step)
    cols = ["x", "y", "z", aux] + ([extra] if extra else [])
    out = np.empty((n, len(cols)), np.float32)
    for k, name in enumerate(cols):
        off, dt, _ = msg["fields"][name]
        if dt != 7:
            raise BagError(f"field {name} is not FLOAT32")
        out[:, k] = np.ascontiguousarray(rec[:, off:off + 4]).view("<f4")[:, 0]
    return out


def decode_imu(raw):
    """sensor_msgs/Imu -> dict(header, orientation (x,y,z,w), angular_velocity (3,), linear_acceleration (3,))."""
    r = _R(raw)
    h = _header(r)
    ori = np.array(r.u("4d")); r.u("9d")
    gyr = np.array(r.u("3d")); r.u("9d")
    acc = np.array(r.u("3d")); r.u("9d")
    return dict(header=h, orientation=ori, angular_velocity=gyr, linear_acceleration=acc)


def decode_livox_custom(raw):
    """livox_ros_driver/CustomMsg -> dict(header, timebase, point_num, lidar_id, points = CUSTOM_POINT array)."""
    r = _R(raw)
    h = _header(r)
    timebase = r.u("Q")
    point_num = r.u("I")
    lidar_id = r.u("B")
    r.raw(3)
    n = r.u("I")
    pts = np.frombuffer(r.raw(n * CUSTOM_POINT.itemsize), CUSTOM_POINT)
    return dict(header=h, timebase=timebase, point_num=point_num, lidar_id=lidar_id, points=pts)


DECODERS = {"sensor_msgs/PointCloud2": decode_pointcloud2, "sensor_msgs/Imu": decode_imu,
            "livox_ros_driver/CustomMsg": decode_livox_custom}


# ---------------------------------------------------------------------------------------------------
# serialisation + a minimal bag writer (test infrastructure for the reader; also handy to export synthetic scans)
# ---------------------------------------------------------------------------------------------------
def _ser_header(seq, stamp, frame_id):
    sec = int(stamp)
    nsec = int(round((stamp - sec) * 1e9))
    if nsec >= 1000000000:
        sec, nsec = sec + 1, nsec - 1000000000
    fid = frame_id.encode()
    return struct.pack("<III", seq, sec, nsec) + struct.pack("<I", len(fid)) + fid


def encode_pointcloud2(points, field_names, stamp, frame_id="lili_om", seq=0, point_step=None, offsets=None):
    """points (n, k) float32; field k at offsets[k] (default 4*k) inside records of point_step bytes (default 4*k)."""
    pts = np.ascontiguousarray(points, np.float32)
    n, k = pts.shape
    offsets = [4 * i for i in range(k)] if offsets is None else list(offsets)
    point_step = max(offsets) + 4 if point_step is None else point_step
    rec = np.zeros((n, point_step), np.uint8)
    for i in range(k):
        rec[:, offsets[i]:offsets[i] + 4] = pts[:, i:i + 1].view(np.uint8)
    b = _ser_header(seq, stamp, frame_id) + struct.pack("<II", 1, n) + struct.pack("<I", k)
    for name, off in zip(field_names, offsets):
        nb = name.encode()
        b += struct.pack("<I", len(nb)) + nb + struct.pack("<IBI", off, 7, 1)
    b += struct.pack("<BII", 0, point_step, point_step * n) + struct.pack("<I", rec.size) + rec.tobytes() + struct.pack("<B", 1)
    return b


def encode_imu(stamp, gyr, acc=(0.0, 0.0, 9.81), frame_id="imu", seq=0):
    z9 = struct.pack("<9d", *([0.0] * 9))
    return (_ser_header(seq, stamp, frame_id) + struct.pack("<4d", 0.0, 0.0, 0.0, 1.0) + z9 +
            struct.pack("<3d", *[float(v) for v in gyr]) + z9 + struct.pack("<3d", *[float(v) for v in acc]) + z9)


def encode_livox_custom(points, stamp, timebase=0, lidar_id=0, frame_id="livox_frame", seq=0):
    pts = np.ascontiguousarray(points, CUSTOM_POINT)
    return (_ser_header(seq, stamp, frame_id) + struct.pack("<QIB3x", int(timebase), pts.shape[0], lidar_id) +
            struct.pack("<I", pts.shape[0]) + pts.tobytes())


def _field(name, value):
    kv = name.encode() + b"=" + value
    return struct.pack("<I", len(kv)) + kv


def _record(fields, data):
    hdr = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", len(data)) + data


def write_bag(path, messages, compression="none", chunk_messages=8):
    """messages: iterable of (topic, msgtype, t_receive, raw_bytes), written in the given order."""
    conns, out = {}, bytearray(MAGIC)
    bag_hdr_data = b" " * 4027          # rosbag pads the bag header record to 4096 bytes; the reader does not care
    out += _record([("op", bytes([OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", 0)),
                    ("chunk_count", struct.pack("<I", 0))], bag_hdr_data)
    chunk = bytearray()
    count = 0

    def flush():
        nonlocal chunk, count
        if not chunk:
            return
        raw = bytes(chunk)
        if compression == "bz2":
            payload = bz2.compress(raw)
        elif compression == "none":
            payload = raw
        else:
            raise BagError("writer supports none / bz2")
        out.extend(_record([("op", bytes([OP_CHUNK])), ("compression", compression.encode()), ("size", struct.pack("<I", len(raw)))], payload))
        chunk, count = bytearray(), 0

    for topic, mtype, t, raw in messages:
        if topic not in conns:
            cid = len(conns)
            conns[topic] = cid
            info = _field("topic", topic.encode()) + _field("type", mtype.encode()) + _field("md5sum", b"0" * 32) + _field("message_definition", b"")
            chunk += _record([("op", bytes([OP_CONNECTION])), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], info)
        sec = int(t)
        nsec = int(round((t - sec) * 1e9))
        chunk += _record([("op", bytes([OP_MSG])), ("conn", struct.pack("<I", conns[topic])), ("time", struct.pack("<II", sec, nsec))], raw)
        count += 1
        if count >= chunk_messages:
            flush()
    flush()
    with open(path, "wb") as f:
        f.write(bytes(out))
