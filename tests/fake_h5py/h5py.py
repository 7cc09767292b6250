"""TEST INFRASTRUCTURE: an in-memory stand-in for the small part of ``h5py`` that the reference's ``tenpy.tools.hdf5_io``
uses (h5py is not installed in this image, there is no network): ``File`` (context manager; the tree is pickled to the file
name on close and unpickled on open, so one process can write and another read), ``Group`` (``create_group``, ``keys``,
item access by '/'-separated paths, hard links, ``attrs``, ``id``, ``name``), ``Dataset`` (``[()]``, ``asstr()``, ``attrs``).
Put ``tests/fake_h5py`` on ``sys.path`` to make ``import h5py`` find it.  Not an HDF5 writer: it pins the LAYOUT (group /
dataset names, attributes, memo links) that ``Array.save_hdf5`` / ``LegCharge.save_hdf5`` produce, which is what a format is."""
import itertools
import pickle

import numpy as np


class _Version:
    version_tuple = (3, 9, 0)
    version = "3.9.0-fake"


version = _Version()
_ids = itertools.count(1)


class _Node:
    def __init__(self, name):
        self.attrs = {}
        self.id = next(_ids)
        self.name = name


class Dataset(_Node):
    def __init__(self, name, value):
        _Node.__init__(self, name)
        if isinstance(value, str):
            self.value = value
        elif isinstance(value, bytes):
            try:
                self.value = value.decode()
            except UnicodeDecodeError:
                self.value = np.array(np.void(value))
        else:
            self.value = np.array(value)        # copies; scalars become 0-d arrays

    @property
    def shape(self):
        return () if isinstance(self.value, str) else self.value.shape

    @property
    def dtype(self):
        return np.dtype('O') if isinstance(self.value, str) else self.value.dtype

    def __getitem__(self, key):
        if isinstance(self.value, str):
            return self.value.encode()          # h5py hands back bytes unless asstr() is used
        if key == ():
            return self.value[()] if self.value.ndim == 0 else self.value.copy()
        return self.value[key]

    def asstr(self):
        ds = self

        class _S:
            def __getitem__(self, key):
                return ds.value
        return _S()


class Group(_Node):
    def __init__(self, name='/', root=None):
        _Node.__init__(self, name)
        self.children = {}
        self._root = root

    @property
    def file(self):
        return self._root if self._root is not None else self

    def __delitem__(self, path):
        parts = [p for p in path.split('/') if p]
        parent = self._walk('/'.join(parts[:-1]))
        del parent.children[parts[-1]]

    def _walk(self, path, create=False):
        node = self
        parts = [p for p in path.split('/') if p]
        for i, p in enumerate(parts):
            if p not in node.children:
                if not create:
                    raise KeyError(path)
                node.children[p] = Group((node.name.rstrip('/') + '/' + p), self.file)
            node = node.children[p]
        return node

    def create_group(self, path):
        parts = [p for p in path.split('/') if p]
        parent = self._walk('/'.join(parts[:-1]), create=True)
        if parts[-1] in parent.children:
            raise ValueError("group exists: " + path)
        g = Group(parent.name.rstrip('/') + '/' + parts[-1], self.file)
        parent.children[parts[-1]] = g
        return g

    def keys(self):
        return list(self.children.keys())

    def __contains__(self, path):
        try:
            self._walk(path)
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        if path in ('/', ''):
            return self
        return self._walk(path)

    def __setitem__(self, path, value):
        parts = [p for p in path.split('/') if p]
        parent = self._walk('/'.join(parts[:-1]), create=True)
        if parts[-1] in parent.children:
            raise ValueError("name exists: " + path)
        if isinstance(value, _Node):
            parent.children[parts[-1]] = value       # hard link (hdf5_io memo)
        else:
            parent.children[parts[-1]] = Dataset(parent.name.rstrip('/') + '/' + parts[-1], value)


class File(Group):
    def __init__(self, filename, mode='r'):
        Group.__init__(self, '/')
        self.filename, self.mode = filename, mode
        self._open = True
        if mode in ('w-', 'x'):
            import os
            if os.path.exists(filename):
                raise FileExistsError(filename)
        if mode in ('r', 'r+', 'a'):
            try:
                with open(filename, 'rb') as f:
                    root = pickle.load(f)
                self.children, self.attrs = root.children, root.attrs
                todo = list(self.children.values())
                while todo:                      # the loaded groups belong to THIS file object
                    g = todo.pop()
                    if isinstance(g, Group):
                        g._root = self
                        todo.extend(g.children.values())
            except FileNotFoundError:
                if mode == 'r':
                    raise

    def __bool__(self):
        return self._open

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        self._open = False
        if self.mode != 'r':
            root = Group('/')
            root.children, root.attrs = self.children, self.attrs
            with open(self.filename, 'wb') as f:
                pickle.dump(root, f)
