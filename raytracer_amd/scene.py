"""Scene container: what the reference's Scene hands to the renderer and to Ads::Build
(src/scene.h:16-56 of the reference): triangles in file order with per-vertex normals and a material
id each, and the material table.  Scenes are stored as .npz dumps of those arrays."""
import numpy as np


class Scene:
    def __init__(self, pos, norm, matid, materials, has_normals=True):
        self.pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3, 3)
        self.norm = None if norm is None or not has_normals else np.ascontiguousarray(norm, np.float32).reshape(-1, 3, 3)
        self.matid = np.ascontiguousarray(matid, np.int32)
        self.materials = np.ascontiguousarray(materials, np.float32).reshape(-1, 18)
        if self.matid.shape[0] != self.pos.shape[0]:
            raise ValueError("one material id per triangle is required")
        if self.matid.size and (self.matid.min() < 0 or self.matid.max() >= self.materials.shape[0]):
            raise ValueError("material id out of range")  # obj_loader.cc:184-188 would index out of range

    @property
    def n_triangles(self):
        return self.pos.shape[0]

    @property
    def has_normals(self):
        return self.norm is not None

    def light_triangles(self):
        """Indices of emissive triangles in file order (obj_loader.cc:188-191)."""
        e = self.materials[self.matid][:, 12:15]
        return np.flatnonzero((e != 0).any(axis=1))

    @classmethod
    def load_npz(cls, path):
        with np.load(path, allow_pickle=False) as z:
            return cls(z["pos"], z["norm"], z["matid"], z["materials"], bool(z["has_normals"]))

    def save_npz(self, path):
        norm = self.norm if self.norm is not None else np.zeros_like(self.pos)
        np.savez_compressed(path, pos=self.pos, norm=norm, matid=self.matid, materials=self.materials,
                            has_normals=np.array(self.has_normals))
