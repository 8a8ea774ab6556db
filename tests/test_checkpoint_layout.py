"""hypad_amd.train._SavedLayout: a module's later checkpoint files written as its first file's archive with the storage record replaced
(what `torch.save` would have written, without pickling the object graph again).  Host logic: runs on the CPU with a stand-in module."""
import io
import zipfile

import torch

from hypad_amd.train import _SavedLayout


class Flat(torch.nn.Module):
    """parameters as views of one flat storage, plus a small constant tensor of its own (like the arena modules' curvature)"""

    def __init__(self, n=20):
        super().__init__()
        self.flat = torch.nn.Parameter(torch.arange(n, dtype=torch.float32))
        self.lin = torch.nn.Linear(4, 3)
        self.lin.weight = torch.nn.Parameter(self.flat.data[:12].view(3, 4))
        self.lin.bias = torch.nn.Parameter(self.flat.data[12:15])
        self.register_buffer("k", torch.tensor([-1.0]))


def _saved(m):
    b = io.BytesIO()
    torch.save(m, b)
    return b.getvalue()


def test_later_files_equal_what_torch_save_would_write():
    m = Flat()
    raw = _saved(m)
    lay = _SavedLayout.parse(raw, m.flat.detach().numpy().tobytes())
    assert lay is not None and lay.nbytes == 80
    for scale in (-2.0, 0.5):
        new = torch.arange(20, dtype=torch.float32) * scale
        out = io.BytesIO()
        lay.write(out, new.numpy().tobytes())
        out.seek(0)
        got = torch.load(out, weights_only=False)
        with torch.no_grad():
            m.flat.copy_(new)
        want = torch.load(io.BytesIO(_saved(m)), weights_only=False)
        assert type(got) is Flat and got.lin.weight.data_ptr() == got.flat.data_ptr()          # the views still share the storage
        for (ka, va), (kb, vb) in zip(got.state_dict().items(), want.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka
        assert torch.equal(got.k, torch.tensor([-1.0]))


def test_layouts_it_does_not_recognise_are_refused():
    m = Flat()
    raw = _saved(m)
    assert _SavedLayout.parse(raw, b"\0" * 80) is None                                          # no record holds these bytes
    assert _SavedLayout.parse(raw, m.flat.detach().numpy().tobytes()[:40]) is None              # nor a record of this size
    two = Flat()
    two.other = torch.nn.Parameter(torch.arange(20, dtype=torch.float32))                      # a second storage with the same bytes: ambiguous
    assert _SavedLayout.parse(_saved(two), two.flat.detach().numpy().tobytes()) is None
    z = io.BytesIO()
    with zipfile.ZipFile(io.BytesIO(raw)) as src, zipfile.ZipFile(z, "w", compression=zipfile.ZIP_DEFLATED) as dst:
        for i in src.infolist():
            dst.writestr(i.filename, src.read(i))
    assert _SavedLayout.parse(z.getvalue(), m.flat.detach().numpy().tobytes()) is None          # compressed members
    assert _SavedLayout.parse(b"not a zip", b"") is None
    lay = _SavedLayout.parse(raw, m.flat.detach().numpy().tobytes())
    try:
        lay.write(io.BytesIO(), b"\0" * 4)
        assert False
    except Exception as e:
        assert "size" in str(e)


def _data_offsets(raw):
    """{member name: file offset of its first data byte} from the local headers"""
    import struct
    out = {}
    with zipfile.ZipFile(io.BytesIO(raw)) as z:
        for i in z.infolist():
            n, e = struct.unpack("<HH", raw[i.header_offset + 26: i.header_offset + 30])
            out[i.filename] = i.header_offset + 30 + n + e
    return out


def test_rewritten_archive_keeps_torch_saves_64_byte_record_alignment():
    m = Flat(n=48)
    m.lin.weight = torch.nn.Parameter(m.flat.data[:12].view(3, 4))
    raw = _saved(m)
    ref = _data_offsets(raw)
    assert all(off % 64 == 0 for name, off in ref.items() if "/data/" in name)                 # what torch.save itself does
    lay = _SavedLayout.parse(raw, m.flat.detach().numpy().tobytes())
    out = io.BytesIO()
    lay.write(out, (m.flat.detach() * 3).numpy().tobytes())
    mine = _data_offsets(out.getvalue())
    assert set(mine) == set(ref)
    assert all(off % 64 == 0 for off in mine.values()), mine
    out.seek(0)
    assert torch.equal(torch.load(out, weights_only=False).flat.detach(), m.flat.detach() * 3)
