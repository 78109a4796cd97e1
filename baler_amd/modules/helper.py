"""Facade of the CLI modes (mirror of ``baler/modules/helper.py`` for train/compress/decompress).

Same function names, argument meaning and return shapes as the reference so ``baler_amd.baler`` reads
like ``baler.baler``; the loops inside ``compress`` / ``decompress`` are replaced by ONE native call
per large row block with a pre-allocated output (the reference grows a numpy array by
``np.concatenate`` per 512-row batch, helper.py:608-611,720-723 -- quadratic copying).
"""
import argparse
import importlib
import os
import sys
from dataclasses import dataclass
from math import ceil

import numpy as np
import torch

sys.path.append(os.getcwd())

from .. import dist as bdist
from .. import hostio
from .. import native
from . import data_processing, training

# rows per native encode/decode call when streaming a dataset through the device
ROW_BLOCK = 1 << 22


def get_arguments(argv=None):
    """reference helper.py:34-101: --mode / --project WORKSPACE PROJECT / --verbose, then
    ``workspaces.<W>.<P>.config.<P>_config.set_config(Config)``."""
    parser = argparse.ArgumentParser(
        prog="baler_amd",
        description="MI355X-native train/compress/decompress path of Baler (same CLI as `baler`).",
        formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument("--mode", type=str, required=True,
                        help="newProject, train, compress, decompress, info")
    parser.add_argument("--project", type=str, required=True, nargs=2, metavar=("WORKSPACE", "PROJECT"),
                        help="Specifies workspace and project, e.g. --project CMS_workspace CMS_project_v1")
    parser.add_argument("--verbose", dest="verbose", action="store_true", help="Verbose mode")
    parser.set_defaults(verbose=False)
    args = parser.parse_args(argv)

    workspace_name, project_name = args.project
    config_path = f"workspaces.{workspace_name}.{project_name}.config.{project_name}_config"
    if args.mode == "newProject":
        config = None
    else:
        config = Config
        importlib.import_module(config_path).set_config(config)
    return config, args.mode, workspace_name, project_name, args.verbose


def create_new_project(workspace_name: str, project_name: str, verbose: bool = False,
                       base_path: str = "workspaces") -> None:
    """reference helper.py:104-147: directory tree + default config."""
    workspace_path = os.path.join(base_path, workspace_name)
    project_path = os.path.join(base_path, workspace_name, project_name)
    if os.path.exists(project_path):
        print(f"The workspace and project ({project_path}) already exists.")
        return
    os.makedirs(project_path)
    required = [
        os.path.join(workspace_path, "data"),
        os.path.join(project_path, "config"),
        os.path.join(project_path, "output", "compressed_output"),
        os.path.join(project_path, "output", "decompressed_output"),
        os.path.join(project_path, "output", "plotting"),
        os.path.join(project_path, "output", "training"),
    ]
    if verbose:
        print(f"Creating project {project_name} in workspace {workspace_name}...")
    for d in required:
        if verbose:
            print(f"Creating directory {d}...")
        os.makedirs(d, exist_ok=True)
    with open(os.path.join(project_path, "config", f"{project_name}_config.py"), "w") as f:
        f.write(create_default_config(workspace_name, project_name))


@dataclass
class Config:
    """Mutable configuration holder: ``set_config(c)`` assigns attributes onto the CLASS, exactly as the
    reference does (helper.py:92-93,150-179), so config modules are interchangeable."""
    input_path: str
    compression_ratio: float
    epochs: int
    early_stopping: bool
    lr_scheduler: bool
    lr_scheduler_patience: int
    min_delta: int
    model_name: str
    custom_norm: bool
    lr: float
    batch_size: int
    test_size: float
    data_dimension: int


def create_default_config(workspace_name: str, project_name: str) -> str:
    """Default project config with the reference's keys (helper.py:182-232)."""
    lines = [
        ("input_path", f'"workspaces/{workspace_name}/data/{project_name}_data.npz"'),
        ("data_dimension", "1"), ("compression_ratio", "2.0"), ("apply_normalization", "True"),
        ("model_name", '"AE"'), ("model_type", '"dense"'), ("epochs", "5"), ("lr", "0.001"),
        ("batch_size", "512"), ("early_stopping", "True"), ("lr_scheduler", "True"),
        ("early_stopping_patience", "100"), ("min_delta", "0"), ("lr_scheduler_patience", "50"),
        ("custom_norm", "False"), ("reg_param", "0.001"), ("RHO", "0.05"), ("test_size", "0"),
        ("extra_compression", "False"), ("intermittent_model_saving", "False"),
        ("intermittent_saving_patience", "100"), ("mse_avg", "False"), ("mse_sum", "True"),
        ("emd", "False"), ("l1", "True"), ("activation_extraction", "False"),
        ("deterministic_algorithm", "True"), ("separate_model_saving", "False"),
        ("save_error_bounded_deltas", "False"), ("error_bounded_requirement", "10"),
        ("convert_to_blocks", "False"),
    ]
    body = "\n".join(f"    c.{k:<30} = {v}" for k, v in lines)
    return f"\n# === Configuration options ===\n\ndef set_config(c):\n{body}\n"


def model_init(model_name: str):
    return data_processing.initialise_model(model_name)


def numpy_to_tensor(data):
    return torch.from_numpy(data)


def get_device():
    """reference helper.py:425-439 returns "cuda:0" or "cpu"; here the device is the rank's GPU and a
    missing GPU is an error (no CPU fallback for the hot path)."""
    native.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def detacher(tensor):
    return tensor.cpu().detach().numpy()


def normalize(data, custom_norm):
    """reference helper.py:261-274 (per-column min-max over axis 0; identity if custom_norm)."""
    return data_processing.normalize(data, custom_norm)


def renormalize(data, true_min_list, feature_range_list, int_mask=None):
    return data_processing.renormalize_func(data, true_min_list, feature_range_list, int_mask)


def npz_array_shape(path, key):
    """Shape of one array of an .npz archive from its .npy header (np.load(path)[key].shape reads the whole array)."""
    import zipfile
    with zipfile.ZipFile(path) as zf, zf.open(key + ".npy") as f:
        version = np.lib.format.read_magic(f)
        header = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        return tuple(header[0])


def _open_table(path, convert_to_blocks=None):
    """-> (host view of the table -- a memory map when the archive stores it uncompressed --, original shape)."""
    src = hostio.open_npz_array(path, "data")
    original_shape = tuple(src.shape)
    if convert_to_blocks:
        print("Converted Dataset to Blocks of Size - ", convert_to_blocks, " from original ", original_shape)
        src = src.reshape(-1, convert_to_blocks[1], convert_to_blocks[2])   # data_processing.py:26-34 (a view)
    return src, original_shape


def _load_to_device(path, plan=None, convert_to_blocks=None):
    """Rows `plan` (default: all) of the archive's table -> device tensor, streamed through pinned staging."""
    src, original_shape = _open_table(path, convert_to_blocks)
    return hostio.upload_rows(src, plan, get_device()), original_shape


def _minmax_features(local, world):
    """find_minmax (data_processing.py:113-130) of a row-sharded table: per-rank column extrema, MIN / MAX all-reduce,
    range = max - min -- bit-identical to the single-process reduction.  -> (2, prod(shape[1:])) float64, device."""
    flat = local.reshape(local.shape[0], -1)
    if world == 1:
        return native.minmax(flat)
    if flat.shape[0] > 0:
        mm = native.col_minmax(flat)
    else:      # a rank without rows: neutral elements
        mm = torch.stack([torch.full((flat.shape[1],), float("inf"), dtype=torch.float64, device=flat.device),
                          torch.full((flat.shape[1],), float("-inf"), dtype=torch.float64, device=flat.device)])
    bdist.allreduce_minmax(mm)
    return torch.stack([mm[0], mm[1] - mm[0]])


def process(input_path, custom_norm, test_size, apply_normalization, convert_to_blocks, verbose, batch_size=None):
    """reference helper.py:277-319.  Returns (train_set, test_set, normalization_features, original_shape); the
    features are a numpy array as in the reference, the two sets live on the DEVICE (the dataset crosses PCIe once,
    through pinned double-buffered staging).

    Under ``torch.distributed`` with ``batch_size`` given (the GLOBAL batch, dist.global_batch) every rank reads from
    the file, uploads and normalises ONLY the rows it trains on -- its slice of every global batch (SURVEY.md section
    8(e)) -- and the sets are ``training.ShardedRows``; the column min/max comes from one MIN/MAX all-reduce."""
    rank, world = bdist.rank_world()
    src, original_shape = _open_table(input_path, convert_to_blocks)
    if verbose:
        print("Original Dataset Shape - ", tuple(original_shape))
    n = src.shape[0]
    sharded = world > 1 and batch_size is not None
    if not sharded:
        data = hostio.upload_rows(src, None, get_device())
        feats = _minmax_features(data, 1).reshape((2,) + tuple(data.shape[1:]))
        train_plan = test_plan = None
    else:
        if not test_size:
            train_plan = test_plan = hostio.RowPlan.cyclic(n, batch_size, rank, world)
            data = hostio.upload_rows(src, train_plan, get_device())
            feats = _minmax_features(data, world)               # the ranks' shards tile the table
        else:
            tr_idx, te_idx = data_processing.split_indices(n, test_size, random_state=1)
            train_plan = hostio.RowPlan.cyclic(n, batch_size, rank, world, index=tr_idx)
            test_plan = hostio.RowPlan.cyclic(n, batch_size, rank, world, index=te_idx)
            # min/max is over the WHOLE table (helper.py:300-303 runs before the split): a contiguous 1/N slice per
            # rank, streamed through the device once and dropped
            probe = hostio.upload_rows(src, hostio.RowPlan.contiguous(n, rank, world), get_device())
            feats = _minmax_features(probe, world)
            del probe
            data = None
        feats = feats.reshape((2,) + tuple(src.shape[1:]))
    normalization_features = feats.cpu().numpy()

    def norm(t):
        if not (apply_normalization and not custom_norm) or t.shape[0] == 0:
            return t
        flat = t.reshape(t.shape[0], -1)
        return native.normalize(flat.contiguous(), feats.reshape(2, -1).contiguous(), torch.float64).reshape(t.shape)

    if apply_normalization:
        print("Normalizing the data...")
    if not sharded:
        data = norm(data)
        if not test_size:
            return data, data, normalization_features, original_shape
        train_set, test_set = data_processing.split(data, test_size=test_size, random_state=1)
        return train_set, test_set, normalization_features, original_shape

    def shard(plan, local):
        return training.ShardedRows(norm(local), plan.n_global, plan.local_spans, batch_size, rank, world)

    if not test_size:
        train_set = shard(train_plan, data)
        return train_set, train_set, normalization_features, original_shape
    train_set = shard(train_plan, hostio.upload_rows(src, train_plan, get_device()))
    test_set = shard(test_plan, hostio.upload_rows(src, test_plan, get_device()))
    return train_set, test_set, normalization_features, original_shape


def train(model, number_of_columns, train_set, test_set, project_path, config):
    return training.train(model, number_of_columns, train_set, test_set, project_path, config)


def model_saver(model, model_path):
    return data_processing.save_model(model, model_path)


def _derive_sizes(config, data_shape, names_len):
    """n_features / latent size exactly as helper.compress derives them (helper.py:505-535)."""
    if config.data_dimension == 1:
        number_of_columns = names_len
        config.latent_space_size = ceil(number_of_columns / config.compression_ratio)
        config.number_of_columns = number_of_columns
        return number_of_columns
    if config.data_dimension == 2:
        if getattr(config, "model_type", None) != "dense":
            raise NotImplementedError("baler_amd covers the dense models; convolutional models are out of scope")
        number_of_rows = data_shape[1]
        config.number_of_columns = data_shape[2]
        config.latent_space_size = ceil((number_of_rows * config.number_of_columns) / config.compression_ratio)
        return number_of_rows * config.number_of_columns
    raise NameError("Data dimension can only be 1 or 2. Got config.data_dimension = "
                    + str(config.data_dimension))


def compress(model_path, config):
    """reference helper.py:473-616.  Returns (compressed ndarray, batches, deltas, indices); the last three are
    empty lists unless ``config.save_error_bounded_deltas`` (then: batch numbers, one float16 array per batch and
    one ``(rows, cols)`` tuple per batch, helper.py:589-606).  The input is re-normalised with ITS OWN min/max
    (helper.py:500-504).  With ``torch.distributed`` every rank reads, uploads and encodes only its contiguous row
    range (no data-path collective; the min/max of the sharded table is one MIN/MAX all-reduce of 2 x C doubles) and
    the latent codes are gathered GPU to GPU onto rank 0, which alone returns them (other ranks: 0 rows).

    Pipeline per rank: file -> pinned staging -> HBM (double buffered), column min/max, then per ROW_BLOCK one
    ``bamd_encode`` with the normalisation fused into the load (evaluated in float64 on the UNCAST source values, as
    the reference normalises before it casts, helper.py:500-504,560-563), the download of block k overlapping the
    encode of block k+1."""
    want_deltas = bool(getattr(config, "save_error_bounded_deltas", False))
    rank, world = bdist.rank_world()
    src, original_shape = _open_table(config.input_path, getattr(config, "convert_to_blocks", None) or None)
    names = np.load(config.input_path)["names"]
    n_features = _derive_sizes(config, src.shape, len(names))
    n_total = src.shape[0]
    plan = hostio.RowPlan.contiguous(n_total, rank, world)
    flat = hostio.upload_rows(src, plan, get_device())
    flat = flat.reshape(flat.shape[0], -1)
    feats = None
    if config.apply_normalization and not config.custom_norm:
        print("Normalizing...")
        feats = _minmax_features(flat, world)
    # reference: 1-D tables stay float64, 2-D ones become float32 AFTER normalisation (helper.py:560-563)
    work_dtype = torch.float32 if config.data_dimension == 2 else flat.dtype
    if feats is None and flat.dtype != work_dtype:
        flat = flat.to(work_dtype)

    model = data_processing.load_model(data_processing.initialise_model(config.model_name), model_path,
                                       n_features=n_features, z_dim=config.latent_space_size)
    model.eval()
    h = model.handle()
    n_local = flat.shape[0]
    out = torch.empty((n_local, config.latent_space_size), dtype=work_dtype, device=flat.device)
    if want_deltas:
        flags = torch.empty((n_local, n_features), dtype=torch.uint8, device=flat.device)
        deltas = torch.empty((n_local, n_features), dtype=torch.float16, device=flat.device)
    ready = []
    for s in range(0, n_local, ROW_BLOCK):
        e = min(s + ROW_BLOCK, n_local)
        if not want_deltas:
            h.encode(flat[s:e], features=feats, out=out[s:e])
        else:
            # the side channel compares decode(encode(x)) with the NORMALISED input (helper.py:589-606)
            xn = native.normalize(flat[s:e], feats, out_dtype=work_dtype) if feats is not None else flat[s:e]
            h.encode(xn, out=out[s:e])
            flags[s:e], deltas[s:e] = native.error_deltas(xn, h.decode(out[s:e]), config.error_bounded_requirement)
        ev = torch.cuda.Event()
        ev.record()
        ready.append((e, ev))
    compressed = _gather_rows(out, n_total, world, ready)
    if not want_deltas:
        return compressed, [], [], []
    flags = _gather_rows(flags, n_total, world)
    deltas = _gather_rows(deltas, n_total, world)
    if rank != 0:
        return compressed, [], [], []
    return (compressed,) + split_deltas(flags, deltas, config.batch_size)


def save_error_bounded_requirement(config, decoded_output, data_batch):
    """reference helper.py:442-470 for ONE batch of host arrays (kept for callers of the reference function; the
    compress path evaluates the whole table at once): -> (list of float16 deltas, (rows, cols))."""
    dev = get_device()
    x = torch.as_tensor(np.ascontiguousarray(data_batch), device=dev)
    r = torch.as_tensor(np.ascontiguousarray(decoded_output), device=dev).to(x.dtype)
    flags, deltas = native.error_deltas(x, r, config.error_bounded_requirement)
    rows, cols = np.nonzero(flags.cpu().numpy())
    return list(deltas.cpu().numpy()[rows, cols]), (rows, cols)


def split_deltas(flags, deltas, batch_size):
    """Dense (flags, float16 deltas) of the whole table -> the reference's per-batch side channel
    (helper.py:589-606): batch numbers, flagged deltas (row-major, as ``np.where`` orders them) and
    ``(rows, cols)`` with rows counted inside the batch.  Every batch is listed, flagged or not."""
    batches, out_deltas, out_index = [], [], []
    for idx, s in enumerate(range(0, flags.shape[0], batch_size)):
        rows, cols = np.nonzero(flags[s:s + batch_size])
        batches.append(idx)
        out_deltas.append(np.ascontiguousarray(deltas[s:s + batch_size][rows, cols]))
        out_index.append((rows, cols))
    print("Total Deltas Found - ", int(sum(len(d) for d in out_deltas)))
    return batches, out_deltas, out_index


def save_deltas(comp_dir, batches, deltas, index):
    """baler.py:316-338: the two gzip'd ``np.save`` object arrays the reference's decompress reads back
    (``compressed_deltas.npz.gz``: one entry per batch; ``compressed_batch_index_metadata.npz.gz``:
    ``[batch numbers, (rows, cols) per batch]``).  Entries are float16 ARRAYS where the reference stored Python
    lists of float16 scalars -- the consumer indexes them the same way (helper.py:708-718)."""
    import gzip
    d_arr = np.empty(len(batches), dtype=object)
    i_arr = np.empty((2, len(batches)), dtype=object)
    for k, b in enumerate(batches):
        d_arr[k] = deltas[k]
        i_arr[0, k] = b
        i_arr[1, k] = index[k]
    with gzip.GzipFile(os.path.join(comp_dir, "compressed_deltas.npz.gz"), "w") as f:
        np.save(file=f, arr=d_arr)
    with gzip.GzipFile(os.path.join(comp_dir, "compressed_batch_index_metadata.npz.gz"), "w") as f:
        np.save(file=f, arr=i_arr)


def load_deltas(input_path_deltas, input_batch_index, batch_size):
    """Side channel files -> flat (global rows int64, cols int32, deltas float16) host arrays (helper.py:655-665)."""
    import gzip
    loaded_deltas = np.load(gzip.GzipFile(input_path_deltas, "r"), allow_pickle=True)
    loaded_index = np.load(gzip.GzipFile(input_batch_index, "r"), allow_pickle=True)
    rows, cols, vals = [], [], []
    for k, b in enumerate(loaded_index[0]):
        r, c = loaded_index[1][k]
        rows.append(np.asarray(r, dtype=np.int64) + int(b) * batch_size)
        cols.append(np.asarray(c, dtype=np.int32))
        vals.append(np.asarray(loaded_deltas[k], dtype=np.float16).reshape(-1))
    if not rows:
        return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float16)
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def _gather_rows(local, n_total, world, ready=None):
    """Per-rank row shards (contiguous ranges in rank order) -> ONE host array on rank 0; other ranks get a 0-row
    array (only rank 0 writes artefacts).  Multi-rank: one device-to-device gather onto rank 0 (dist.gather_rows, RCCL
    over xGMI), then a single pinned, double-buffered download -- nothing is pickled and no rank but 0 touches PCIe.
    ``ready``: per-block completion events of the producer (single rank: the download of block k overlaps block k+1)."""
    if world == 1:
        return hostio.download_rows(local, ready=ready)
    full = bdist.gather_rows(local, n_total, dst=0)
    if full is None:
        return np.empty((0,) + tuple(local.shape[1:]), dtype=hostio.NP_OF_TORCH[local.dtype])
    return hostio.download_rows(full)


def decompress(model_path, input_path, input_path_deltas, input_batch_index, model_name, config,
               output_path, original_shape, renorm=None):
    """reference helper.py:619-733.  Returns (decompressed ndarray, names, normalization_features).  With
    ``config.save_error_bounded_deltas`` the stored float16 deltas are subtracted from the decoder output of
    their batch (helper.py:708-718), on the device, before un-normalisation.

    ``renorm`` (not in the reference signature; optional): ``(features (2, C) float64, int_mask or None)``.  When
    given, the un-normalisation ``x*range + min`` and the truncation of the "int" columns that the reference applies
    afterwards on the host (baler.py:410-435) run on the device -- fused into the decode kernel's store when there are
    no deltas -- so the decompressed table crosses PCIe once (float64) instead of three times."""
    loaded = np.load(input_path)
    data = hostio.open_npz_array(input_path, "data")     # memory-mapped when stored: a rank reads only its rows
    names = loaded["names"]
    normalization_features = loaded["normalization_features"]
    latent_space_size = data.shape[1]
    model_dict = torch.load(str(model_path), map_location="cpu")
    number_of_columns = len(model_dict[list(model_dict.keys())[-1]])  # len(de4.bias), helper.py:668-674

    model = data_processing.load_model(data_processing.initialise_model(config.model_name), model_path,
                                       n_features=number_of_columns, z_dim=latent_space_size)
    model.eval()
    h = model.handle()
    rank, world = bdist.rank_world()
    n_total = data.shape[0]
    lo, hi = bdist.shard_rows(n_total, rank, world)
    # this rank's latent rows only, through pinned double-buffered staging
    z = hostio.upload_rows(data, hostio.RowPlan.contiguous(n_total, rank, world), get_device())
    want_deltas = bool(getattr(config, "save_error_bounded_deltas", False))
    r_feats = r_mask = None
    if renorm is not None:
        r_feats = torch.as_tensor(np.asarray(renorm[0], dtype=np.float64).reshape(2, -1), device=z.device).contiguous()
        if renorm[1] is not None:
            r_mask = torch.as_tensor(np.asarray(renorm[1], dtype=np.uint8), device=z.device).contiguous()
    fuse = renorm is not None and not want_deltas
    out = torch.empty((hi - lo, number_of_columns), dtype=torch.float64 if fuse else z.dtype, device=z.device)
    ready = []
    for s in range(0, hi - lo, ROW_BLOCK):
        e = min(s + ROW_BLOCK, hi - lo)
        if fuse:
            h.decode(z[s:e], features=r_feats, int_mask=r_mask, out=out[s:e])
        else:
            h.decode(z[s:e], out=out[s:e])
        ev = torch.cuda.Event()
        ev.record()
        ready.append((e, ev))
    if want_deltas:
        ready = None
        rows, cols, vals = load_deltas(input_path_deltas, input_batch_index, config.batch_size)
        mine = (rows >= lo) & (rows < hi)                     # this rank's row shard
        dev = out.device
        native.apply_deltas(out, torch.from_numpy(rows[mine] - lo).to(dev), torch.from_numpy(cols[mine]).to(dev),
                            torch.from_numpy(vals[mine]).to(dev))
        print("Total Deltas Added - ", int(len(rows)))
        if renorm is not None:
            out = native.renormalize(out, r_feats, r_mask)
    decompressed = _gather_rows(out, n_total, world, ready)
    if rank != 0:
        return decompressed, names, normalization_features
    if config.data_dimension == 2 and getattr(config, "model_type", None) == "dense":
        blocks = getattr(config, "convert_to_blocks", None)
        if blocks:
            # rows are blocks here, not frames.  The reference reshapes with the FRAME shape (helper.py:728-731) and so
            # raises ValueError for its own exafel1/exafel2 configs (dense model + convert_to_blocks); the caller
            # (perform_decompression, baler.py:394-408) folds the blocks back into frames.
            decompressed = decompressed.reshape((len(decompressed), blocks[1], blocks[2]))
        else:
            decompressed = decompressed.reshape((len(decompressed), original_shape[1], original_shape[2]))
    return decompressed, names, normalization_features
