"""A tiny stand-in for ``h5py.File`` (h5py is not installed in this image) so the Fast5 loaders can be exercised:
files on disk are JSON documents ``{"attrs": {group_path: {name: value}}, "datasets": {path: text}}``; string values
come back as bytes, as h5py returns fixed-length HDF5 strings."""
import json


class _Node:
    def __init__(self, attrs, value):
        self.attrs = {k: (v.encode() if isinstance(v, str) else v) for k, v in attrs.items()}
        self._value = value

    def __getitem__(self, key):
        if key != () or self._value is None:
            raise KeyError(key)
        return self._value.encode()


class FakeH5:
    def __init__(self, path, mode='r'):
        with open(path) as fh:
            doc = json.load(fh)
        self._attrs, self._data = doc.get("attrs", {}), doc.get("datasets", {})
        names = set()
        for p in list(self._attrs) + list(self._data):
            parts = p.split('/')
            for i in range(1, len(parts) + 1):
                names.add('/'.join(parts[:i]))
        self._names = sorted(names)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def visit(self, cb):
        for n in self._names:
            cb(n)

    def keys(self):
        return sorted({n.split('/')[0] for n in self._names})

    def __getitem__(self, path):
        if path not in self._names:
            raise KeyError(path)
        return _Node(self._attrs.get(path, {}), self._data.get(path))

    def flush(self):
        pass


def write_single(path, read_number, channel, mux, stamp, exp_start, fastq=None, summary=None):
    """summary: (sequence_length, mean_qscore) as albacore writes them beside the record (None: the group without them)."""
    attrs = {"Raw/Reads/Read_%d" % read_number: {"start_mux": mux, "read_id": "id%d" % read_number},
             "UniqueGlobalKey/channel_id": {"channel_number": str(channel)},
             "UniqueGlobalKey/tracking_id": {"exp_start_time": exp_start},
             "Analyses/Segmentation_000": {"time_stamp": stamp}}
    data = {}
    if fastq is not None:
        attrs["Analyses/Basecall_1D_000/Summary/basecall_1d_template"] = \
            {} if summary is None else {"sequence_length": int(summary[0]), "mean_qscore": float(summary[1])}
        data["Analyses/Basecall_1D_000/BaseCalled_template/Fastq"] = fastq
    with open(path, "w") as fh:
        json.dump({"attrs": attrs, "datasets": data}, fh)


def write_multi(path, reads, exp_start):
    """reads: list of (name, channel, mux, stamp, fastq_or_None[, (sequence_length, mean_qscore)])."""
    attrs, data = {}, {}
    for rd in reads:
        name, channel, mux, stamp, fastq = rd[:5]
        if len(rd) > 5 and rd[5] is not None:
            attrs[name + "/Analyses/Basecall_1D_000/Summary/basecall_1d_template"] = \
                {"sequence_length": int(rd[5][0]), "mean_qscore": float(rd[5][1])}
        attrs[name + "/Raw"] = {"start_mux": mux}
        attrs[name + "/channel_id"] = {"channel_number": str(channel)}
        attrs[name + "/tracking_id"] = {"exp_start_time": exp_start}
        attrs[name + "/Analyses/Segmentation_000"] = {"time_stamp": stamp}
        if fastq is not None:
            data[name + "/Analyses/Basecall_1D_000/BaseCalled_template/Fastq"] = fastq
    with open(path, "w") as fh:
        json.dump({"attrs": attrs, "datasets": data}, fh)
