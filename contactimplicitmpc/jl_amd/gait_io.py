"""Reader for the reference's gait files (the data format on the input side of the path).

The reference stores its reference trajectories as JLD2 files (`get_trajectory`, src/controller/trajectory.jl:152-184,
`JLD2.jldopen(gait_path)`): an HDF5 container (superblock version 2 behind a 512-byte text header) whose root group
links `qm, um, γm, bm, ψm, ηm, μm, hm` (`load_type = :split_traj_alt`, trajectory.jl:169-170).  A
`Vector{Vector{Float64}}` is a dataset of object references, each pointing at a small Float64 dataset.

This is a from-the-format reader for exactly that subset (object header version 2, link / dataspace / datatype /
data-layout / continuation messages, compact and contiguous layouts, Float64 / Int64 / object-reference element
types).  No Julia and no HDF5 library are needed.  Anything outside the subset raises `GaitFormatError` - the reader
never guesses.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

JLD2_HEADER = 512          # JLD2 prepends a text header; every address in the file is relative to this base


class GaitFormatError(ValueError):
    pass


@dataclass
class _Dataset:
    shape: tuple
    kind: str              # "f8", "i8", "ref"
    raw: bytes


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        if not buf.startswith(b"HDF5-based Julia Data Format"):
            raise GaitFormatError("not a JLD2 file")
        sb = JLD2_HEADER
        if buf[sb:sb + 8] != b"\x89HDF\r\n\x1a\n":
            raise GaitFormatError("HDF5 signature not found behind the JLD2 header")
        version, so, sl = buf[sb + 8], buf[sb + 9], buf[sb + 10]
        if version != 2 or so != 8 or sl != 8:
            raise GaitFormatError(f"unsupported superblock (version {version}, offsets {so}, lengths {sl})")
        self.base, _ext, _eof, self.root = struct.unpack_from("<QQQQ", buf, sb + 12)

    def _abs(self, rel: int) -> int:
        return rel + self.base

    # -- object headers (version 2) ------------------------------------------------------------------
    def messages(self, rel: int):
        """Yields (type, payload bytes) of the object header at the relative address `rel`."""
        b = self.b
        at = self._abs(rel)
        if b[at:at + 4] != b"OHDR" or b[at + 4] != 2:
            raise GaitFormatError(f"no version-2 object header at {at}")
        flags = b[at + 5]
        p = at + 6
        if flags & 0x20:
            p += 16                                   # access / modification / change / birth times
        if flags & 0x10:
            p += 4                                    # max compact / min dense attribute counts
        width = 1 << (flags & 3)
        size = int.from_bytes(b[p:p + width], "little")
        p += width
        chunks = [(p, p + size)]
        order = 2 if flags & 0x04 else 0
        while chunks:
            p, end = chunks.pop(0)
            while p + 4 + order <= end:
                mtype = b[p]
                msize = struct.unpack_from("<H", b, p + 1)[0]
                body = p + 4 + order
                if body + msize > end:
                    break                             # gap at the end of a chunk
                payload = b[body:body + msize]
                if mtype == 0x10:                     # continuation: another chunk ("OCHK" + messages + checksum)
                    off, length = struct.unpack_from("<QQ", payload, 0)
                    a = self._abs(off)
                    if b[a:a + 4] != b"OCHK":
                        raise GaitFormatError("bad continuation chunk")
                    chunks.append((a + 4, a + length - 4))
                elif mtype != 0:
                    yield mtype, payload
                p = body + msize

    def links(self, rel: int) -> dict:
        out = {}
        for mtype, m in self.messages(rel):
            if mtype != 0x06:
                continue
            if m[0] != 1:
                raise GaitFormatError("unsupported link message version")
            flags = m[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = m[p]; p += 1
            if flags & 0x04:
                p += 8                                # creation order
            if flags & 0x10:
                p += 1                                # character set
            w = 1 << (flags & 3)
            n = int.from_bytes(m[p:p + w], "little"); p += w
            name = m[p:p + n].decode("utf-8"); p += n
            if ltype != 0:
                raise GaitFormatError(f"link {name!r}: only hard links are supported")
            out[name] = struct.unpack_from("<Q", m, p)[0]
        return out

    def dataset(self, rel: int) -> _Dataset:
        shape = None
        kind = None
        raw = None
        elsize = 0
        for mtype, m in self.messages(rel):
            if mtype == 0x01:                          # dataspace
                if m[0] != 2:
                    raise GaitFormatError("unsupported dataspace version")
                rank, flags, stype = m[1], m[2], m[3]
                if stype == 0:
                    shape = ()
                elif stype == 1:
                    shape = struct.unpack_from("<" + "Q" * rank, m, 4)
                else:
                    shape = (0,)
            elif mtype == 0x03:                        # datatype
                cls = m[0] & 0x0F
                elsize = struct.unpack_from("<I", m, 4)[0]
                if cls == 1 and elsize == 8:
                    kind = "f8"
                elif cls == 0 and elsize == 8:
                    kind = "i8"
                elif cls == 7 and elsize == 8:
                    kind = "ref"
                else:
                    kind = f"class{cls}/{elsize}"
            elif mtype == 0x08:                        # data layout
                ver, cls = m[0], m[1]
                if ver not in (3, 4):
                    raise GaitFormatError("unsupported layout version")
                if cls == 0:
                    n = struct.unpack_from("<H", m, 2)[0]
                    raw = m[4:4 + n]
                elif cls == 1:
                    addr, n = struct.unpack_from("<QQ", m, 2)
                    a = self._abs(addr)
                    raw = self.b[a:a + n]
                else:
                    raise GaitFormatError("chunked layouts are not supported")
        if shape is None or kind is None or raw is None:
            raise GaitFormatError(f"object at {rel} is not a plain dataset")
        return _Dataset(tuple(shape), kind, raw)

    def value(self, rel: int):
        d = self.dataset(rel)
        n = int(np.prod(d.shape)) if d.shape else 1
        if d.kind == "f8":
            a = np.frombuffer(d.raw, dtype="<f8", count=n).copy()
        elif d.kind == "i8":
            a = np.frombuffer(d.raw, dtype="<i8", count=n).copy()
        elif d.kind == "ref":
            refs = np.frombuffer(d.raw, dtype="<u8", count=n)
            return [self.value(int(r)) for r in refs]
        else:
            raise GaitFormatError(f"unsupported element type {d.kind}")
        if d.shape == ():
            return a[0]
        return a.reshape(d.shape[::-1]).T if len(d.shape) > 1 else a     # HDF5 stores Julia's dims reversed


    def struct_slots(self, rel: int) -> bytes:
        """Raw bytes of a scalar dataset stored with a compact layout, whatever its (committed) datatype - a Julia
        struct saved by JLD2: 8-byte slots, plain fields inline, array fields as relative object addresses."""
        for mtype, m in self.messages(rel):
            if mtype == 0x08 and m[0] in (3, 4) and m[1] == 0:
                n = struct.unpack_from("<H", m, 2)[0]
                return m[4:4 + n]
        raise GaitFormatError(f"object at {rel} has no compact data")


def read_jld2(path) -> dict:
    """All root-level datasets of a gait file: name -> float / ndarray / list of ndarrays."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    out = {}
    for name, rel in r.links(r.root).items():
        try:
            out[name] = r.value(rel)
        except GaitFormatError:
            continue                                   # e.g. the `_types` group JLD2 adds for struct layouts
    return out


@dataclass
class Gait:
    """The fields `get_trajectory(...; load_type = :split_traj_alt)` reads (trajectory.jl:169-170)."""
    q: np.ndarray          # (H + 2, nq)
    u: np.ndarray          # (H, nu)
    gamma: np.ndarray      # (H, nc)
    b: np.ndarray          # (H, nb)
    psi: np.ndarray        # (H, nc)
    eta: np.ndarray        # (H, nb)
    mu: float
    h: float

    @property
    def H(self) -> int:
        return self.u.shape[0]


def load_gait(path) -> Gait:
    d = read_jld2(path)
    need = ["qm", "um", "γm", "bm", "ψm", "ηm", "μm", "hm"]
    missing = [k for k in need if k not in d]
    if missing:
        raise GaitFormatError(f"{path}: not a :split_traj_alt gait file (missing {missing})")
    mu = d["μm"]
    mu = float(np.asarray(mu).reshape(-1)[0])
    return Gait(q=np.stack(d["qm"]), u=np.stack(d["um"]), gamma=np.stack(d["γm"]), b=np.stack(d["bm"]),
                psi=np.stack(d["ψm"]), eta=np.stack(d["ηm"]), mu=mu, h=float(d["hm"]))


@dataclass
class JointTraj:
    """A serialized `ContactTraj` (`load_type = :joint_traj`, trajectory.jl:181-182; fields trajectory.jl:1-19)."""
    H: int
    h: float
    kappa: float
    q: np.ndarray          # (H + 2, nq)
    u: np.ndarray          # (H, nu)
    w: np.ndarray          # (H, nw)
    gamma: np.ndarray      # (H, nc)
    b: np.ndarray          # (H, nb)
    z: np.ndarray          # (H, nz)
    theta: np.ndarray      # (H, nθ)


def load_joint_traj(path) -> JointTraj:
    """Reads the `traj` struct of a :joint_traj gait file (e.g. src/dynamics/hopper_2D/gaits/gait_forward.jld2): 17
    eight-byte slots in the field order of `ContactTraj` - H, h inline, then the addresses of κ, q, u, w, γ, b, z, θ and of
    the seven index vectors.  The shapes are cross-checked against each other; anything unexpected raises."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    links = r.links(r.root)
    if "traj" not in links:
        raise GaitFormatError(f"{path}: no `traj` object (not a :joint_traj gait file)")
    raw = r.struct_slots(links["traj"])
    if len(raw) != 17 * 8:
        raise GaitFormatError(f"{path}: ContactTraj struct of {len(raw)} bytes, expected 136")
    H = struct.unpack_from("<q", raw, 0)[0]
    h = struct.unpack_from("<d", raw, 8)[0]
    fields = [r.value(struct.unpack_from("<Q", raw, 8 * k)[0]) for k in range(2, 17)]
    kappa, q, u, w, g, b, z, th = fields[:8]
    arr = lambda v: np.stack(v) if isinstance(v, list) else np.asarray(v)
    q, u, w, g, b, z, th = map(arr, (q, u, w, g, b, z, th))
    nq, nu, nw, nc, nb = q.shape[1], u.shape[1], w.shape[1], g.shape[1], b.shape[1]
    ok = (q.shape[0] == H + 2 and all(a.shape[0] == H for a in (u, w, g, b, z, th)) and 0.0 < h < 1.0
          and z.shape[1] == nq + 4 * nc + 2 * nb and th.shape[1] == 2 * nq + nu + nw + 2
          and [len(np.atleast_1d(f)) for f in fields[8:]] == [nq, nq, nu, nw, nq, nc, nb])
    if not ok:
        raise GaitFormatError(f"{path}: the struct does not have the ContactTraj field layout")
    return JointTraj(int(H), float(h), float(np.atleast_1d(kappa)[0]), q, u, w, g, b, z, th)
