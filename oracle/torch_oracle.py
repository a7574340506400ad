"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Pure-PyTorch, CPU, fp32, functional restatement of DKT-Stereo's stereo
inference path, written against a *state dict* (plain name -> tensor map) so
that it shares no module code with dkt_stereo_amd/.  It executes the same
primitive sequence as the reference (einsum, avg_pool2d, grid_sample, conv2d,
sigmoid/tanh ...) and is therefore the "port" CPU baseline that bench.py times
(cpu_baseline.kind == "port") and the checker the -m gpu parity tests compare
the HIP path with.

Pinning: tests/golden/make_golden.py imports the reference from
/root/reference (build container only), runs both on identical seeded inputs
and weights, asserts agreement (0.0 max-abs for every kernel-level function,
see tests/golden/MANIFEST.json for the recorded numbers) and stores the
reference's outputs as .npz fixtures; tests/test_oracle.py re-checks this
module against those fixtures everywhere (GPU box included, where the
reference does not exist).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Reference citations are file:line relative to the
reference tree.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# sampling primitive: core/utils/utils.py:59-74 (bilinear_sampler, H == 1 case)
# --------------------------------------------------------------------------
def _sample_rows(rows, x):
    """rows (N,Cv,1,Wd), x (N,1,K,1) pixel coords -> (N,Cv,1,K)."""
    wd = rows.shape[-1]
    xg = 2 * x / (wd - 1) - 1
    grid = torch.cat([xg, torch.zeros_like(xg)], dim=-1)
    return F.grid_sample(rows, grid, align_corners=True)


def _taps(radius, like):
    return torch.linspace(-radius, radius, 2 * radius + 1).to(like.device)


# --------------------------------------------------------------------------
# RAFT-Stereo correlation: core/corr.py:110-156
# --------------------------------------------------------------------------
def corr1d_volume(fmap1, fmap2, scaled=True):
    """core/corr.py:148-156 (scaled) / igev_stereo/geometry.py:62-69 (unscaled)."""
    b, c, h, w1 = fmap1.shape
    w2 = fmap2.shape[3]
    vol = torch.einsum('aijk,aijh->ajkh', fmap1, fmap2).reshape(b, h, w1, 1, w2).contiguous()
    if scaled:
        vol = vol / torch.sqrt(torch.tensor(c).float())
    return vol


def corr1d_pyramid(fmap1, fmap2, num_levels):
    """core/corr.py:111-125 -> the `num_levels` levels __call__ reads, each (N,1,1,W2_i)."""
    vol = corr1d_volume(fmap1, fmap2)
    b, h, w1, _, w2 = vol.shape
    lvl = vol.reshape(b * h * w1, 1, 1, w2)
    pyr = [lvl]
    for _ in range(num_levels - 1):
        lvl = F.avg_pool2d(lvl, [1, 2], stride=[1, 2])
        pyr.append(lvl)
    return pyr


def corr1d_lookup(pyr, coords, radius):
    """core/corr.py:127-146.  coords (B,2,H,W) -> (B, L*K, H, W)."""
    b, _, h, w = coords.shape
    cx = coords[:, :1].permute(0, 2, 3, 1).reshape(b * h * w, 1, 1, 1)
    dx = _taps(radius, coords).view(2 * radius + 1, 1)
    outs = []
    for i, lvl in enumerate(pyr):
        x0 = dx + cx / 2 ** i
        outs.append(_sample_rows(lvl, x0).view(b, h, w, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def corr1d_lookup_alt(fmap1, fmap2, coords, num_levels, radius):
    """core/corr.py:64-107 (PytorchAlternateCorrBlock1D, on-the-fly)."""
    b, c, h, w = fmap1.shape
    xy = coords.permute(0, 2, 3, 1)
    outs = []
    f2 = fmap2
    dx = _taps(radius, coords)
    for i in range(num_levels):
        hh, ww = f2.shape[2:]
        per_tap = []
        for k in range(2 * radius + 1):
            x = xy[..., 0] / 2 ** i + dx[k]
            xg = 2 * x / (ww - 1) - 1
            yg = 2 * xy[..., 1] / (hh - 1) - 1
            g = torch.stack([xg, yg], dim=-1)
            warped = F.grid_sample(f2, g, align_corners=True)
            per_tap.append(torch.sum(warped * fmap1, dim=1))
        outs.append(torch.stack(per_tap, dim=1).permute(0, 2, 3, 1) / torch.sqrt(torch.tensor(c).float()))
        f2 = F.avg_pool2d(f2, [1, 2], stride=[1, 2])
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def corr1d_volume_cosine(fmap1, fmap2):
    """core/corr.py:196-209 (CorrBlock1D_Cosine.corr)."""
    fmap1 = fmap1 / fmap1.norm(dim=1, keepdim=True)
    fmap2 = fmap2 / fmap2.norm(dim=1, keepdim=True)
    return corr1d_volume(fmap1, fmap2, scaled=False)


# --------------------------------------------------------------------------
# IGEV combined geometry encoding volume: meta_arch/igev_stereo/geometry.py:6-58
# --------------------------------------------------------------------------
def geo_pyramids(fmap1, fmap2, geo_volume, num_levels):
    b, c, d, h, w = geo_volume.shape
    init = corr1d_volume(fmap1, fmap2, scaled=False)
    w2 = init.shape[-1]
    geo = geo_volume.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, 1, d)
    init = init.reshape(b * h * w, 1, 1, w2)
    gp, ip = [geo], [init]
    for _ in range(num_levels - 1):
        geo = F.avg_pool2d(geo, [1, 2], stride=[1, 2])
        gp.append(geo)
    for _ in range(num_levels - 1):
        init = F.avg_pool2d(init, [1, 2], stride=[1, 2])
        ip.append(init)
    return gp, ip


def geo_lookup(geo_pyr, init_pyr, disp, coords, radius):
    """geometry.py:34-58.  disp (B,1,H,W), coords (B,H,W,1) -> (B, L*K*(C+1), H, W)."""
    b, _, h, w = disp.shape
    dx = _taps(radius, disp).view(1, 1, 2 * radius + 1, 1)
    dn = disp.reshape(b * h * w, 1, 1, 1)
    cn = coords.reshape(b * h * w, 1, 1, 1)
    outs = []
    for i in range(len(geo_pyr)):
        g = _sample_rows(geo_pyr[i], dx + dn / 2 ** i).view(b, h, w, -1)
        x_init = cn / 2 ** i - dn / 2 ** i + dx
        c0 = _sample_rows(init_pyr[i], x_init).view(b, h, w, -1)
        outs += [g, c0]
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# --------------------------------------------------------------------------
# cost-volume builders
# --------------------------------------------------------------------------
def gwc_volume(ref, tgt, maxdisp, num_groups):
    """igev_stereo/submodule.py:152-170 == gwcnet/submodules.py:39-58."""
    b, c, h, w = ref.shape
    cpg = c // num_groups
    vol = ref.new_zeros([b, num_groups, maxdisp, h, w])
    for d in range(maxdisp):
        if d == 0:
            vol[:, :, 0] = (ref * tgt).view(b, num_groups, cpg, h, w).mean(dim=2)
        elif d < w:  # for d >= w the reference's slices are empty and the plane stays 0
            prod = ref[:, :, :, d:] * tgt[:, :, :, :-d]
            vol[:, :, d, :, d:] = prod.view(b, num_groups, cpg, h, w - d).mean(dim=2)
    return vol.contiguous()


def concat_volume(ref, tgt, maxdisp, ref_masked):
    """ref_masked=True: gwcnet/submodules.py:25-36; False: igev_stereo/submodule.py:207-218."""
    b, c, h, w = ref.shape
    vol = ref.new_zeros([b, 2 * c, maxdisp, h, w])
    for d in range(maxdisp):
        if d == 0:
            vol[:, :c, 0] = ref
            vol[:, c:, 0] = tgt
        else:
            if ref_masked:
                vol[:, :c, d, :, d:] = ref[:, :, :, d:]
            else:
                vol[:, :c, d, :, :] = ref
            if d < w:
                vol[:, c:, d, :, d:] = tgt[:, :, :, :-d]
    return vol.contiguous()


# --------------------------------------------------------------------------
# update operator: core/update.py (RAFT) and meta_arch/igev_stereo/update.py
# --------------------------------------------------------------------------
def _conv(sd, name, x, stride=1):
    w = sd[name + '.weight']
    b = sd.get(name + '.bias')
    return F.conv2d(x, w, b, stride=stride, padding=(w.shape[2] // 2, w.shape[3] // 2))


def conv_gru(sd, pre, h, cz, cr, cq, *x_list):
    """core/update.py:23-32."""
    x = torch.cat(x_list, dim=1)
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(_conv(sd, pre + '.convz', hx) + cz)
    r = torch.sigmoid(_conv(sd, pre + '.convr', hx) + cr)
    q = torch.tanh(_conv(sd, pre + '.convq', torch.cat([r * h, x], dim=1)) + cq)
    return (1 - z) * h + z * q


def motion_encoder(sd, pre, flow, corr, igev=False):
    """core/update.py:77-85; igev_stereo/update.py:84-92 (convd* instead of convf*)."""
    f1, f2 = ('.convd1', '.convd2') if igev else ('.convf1', '.convf2')
    cor = F.relu(_conv(sd, pre + '.convc1', corr))
    cor = F.relu(_conv(sd, pre + '.convc2', cor))
    flo = F.relu(_conv(sd, pre + f1, flow))
    flo = F.relu(_conv(sd, pre + f2, flo))
    out = F.relu(_conv(sd, pre + '.conv', torch.cat([cor, flo], dim=1)))
    return torch.cat([out, flow], dim=1)


def pool2x(x):
    """core/update.py:87-88."""
    return F.avg_pool2d(x, 3, stride=2, padding=1)


def interp(x, dest):
    """core/update.py:93-95."""
    return F.interpolate(x, dest.shape[2:], mode='bilinear', align_corners=True)


def update_block(sd, pre, n_gru_layers, net, inp, corr=None, flow=None,
                 it_fine=True, it_mid=True, it_coarse=True, update=True, igev=False):
    """core/update.py:115-138 (RAFT: gru08/gru16/gru32, flow_head, mask) and
    igev_stereo/update.py:121-142 (gru04/gru08/gru16, disp_head, mask_feat_4).
    `net` is updated in place like the reference does."""
    fine, mid, coarse = ('gru04', 'gru08', 'gru16') if igev else ('gru08', 'gru16', 'gru32')
    if it_coarse:
        net[2] = conv_gru(sd, pre + '.' + coarse, net[2], *inp[2], pool2x(net[1]))
    if it_mid:
        if n_gru_layers > 2:
            net[1] = conv_gru(sd, pre + '.' + mid, net[1], *inp[1], pool2x(net[0]), interp(net[2], net[1]))
        else:
            net[1] = conv_gru(sd, pre + '.' + mid, net[1], *inp[1], pool2x(net[0]))
    if it_fine:
        mf = motion_encoder(sd, pre + '.encoder', flow, corr, igev=igev)
        if n_gru_layers > 1:
            net[0] = conv_gru(sd, pre + '.' + fine, net[0], *inp[0], mf, interp(net[1], net[0]))
        else:
            net[0] = conv_gru(sd, pre + '.' + fine, net[0], *inp[0], mf)
    if not update:
        return net
    head = '.disp_head' if igev else '.flow_head'
    delta = _conv(sd, pre + head + '.conv2', F.relu(_conv(sd, pre + head + '.conv1', net[0])))
    if igev:
        mask = F.relu(_conv(sd, pre + '.mask_feat_4.0', net[0]))
    else:
        mask = .25 * _conv(sd, pre + '.mask.2', F.relu(_conv(sd, pre + '.mask.0', net[0])))
    return net, mask, delta


# --------------------------------------------------------------------------
# encoders (black box for the hot path, needed for end-to-end fixtures):
# core/extractor.py:6-60 (ResidualBlock), :122-197 (BasicEncoder),
# :199-300 (MultiBasicEncoder).  Inference semantics: BatchNorm uses running
# statistics, InstanceNorm2d is affine-free and always uses instance statistics.
# --------------------------------------------------------------------------
def _norm(sd, name, x, kind):
    if kind == 'instance':
        return F.instance_norm(x)
    if kind == 'batch':
        return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                            sd[name + '.weight'], sd[name + '.bias'], False, 0.1, 1e-5)
    if kind == 'none':
        return x
    raise ValueError(kind)


def _res_block(sd, pre, x, kind, stride):
    y = F.relu(_norm(sd, pre + '.norm1', _conv(sd, pre + '.conv1', x, stride), kind))
    y = F.relu(_norm(sd, pre + '.norm2', _conv(sd, pre + '.conv2', y), kind))
    if (pre + '.downsample.0.weight') in sd:
        x = _norm(sd, pre + '.norm3', _conv(sd, pre + '.downsample.0', x, stride), kind)
    return F.relu(x + y)


def _layer(sd, pre, x, kind, stride):
    x = _res_block(sd, pre + '.0', x, kind, stride)
    return _res_block(sd, pre + '.1', x, kind, 1)


def _trunk(sd, pre, x, kind, downsample):
    x = F.relu(_norm(sd, pre + '.norm1', _conv(sd, pre + '.conv1', x, 1 + (downsample > 2)), kind))
    x = _layer(sd, pre + '.layer1', x, kind, 1)
    x = _layer(sd, pre + '.layer2', x, kind, 1 + (downsample > 1))
    x = _layer(sd, pre + '.layer3', x, kind, 1 + (downsample > 0))
    return x


def basic_encoder(sd, pre, x, kind='instance', downsample=2):
    return _conv(sd, pre + '.conv2', _trunk(sd, pre, x, kind, downsample))


def multi_encoder(sd, pre, x, kind='batch', downsample=2, num_layers=3, n_heads=2):
    x = _trunk(sd, pre, x, kind, downsample)
    o08 = [_conv(sd, '%s.outputs08.%d.1' % (pre, j), _res_block(sd, '%s.outputs08.%d.0' % (pre, j), x, kind, 1))
           for j in range(n_heads)]
    if num_layers == 1:
        return (o08,)
    y = _layer(sd, pre + '.layer4', x, kind, 2)
    o16 = [_conv(sd, '%s.outputs16.%d.1' % (pre, j), _res_block(sd, '%s.outputs16.%d.0' % (pre, j), y, kind, 1))
           for j in range(n_heads)]
    if num_layers == 2:
        return (o08, o16)
    z = _layer(sd, pre + '.layer5', y, kind, 2)
    o32 = [_conv(sd, '%s.outputs32.%d' % (pre, j), z) for j in range(n_heads)]
    return (o08, o16, o32)


# --------------------------------------------------------------------------
# RAFT-Stereo forward, test_mode: meta_arch/raft_stereo/raft_stereo.py:85-183
# --------------------------------------------------------------------------
def coords_grid(b, h, w):
    """core/utils/utils.py:77-80."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    return torch.stack([xs, ys], dim=0).float()[None].repeat(b, 1, 1, 1)


def convex_upsample(flow, mask, factor):
    """raft_stereo.py:70-82."""
    n, d, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, factor, factor, h, w), dim=2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(n, d, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, d, factor * h, factor * w)


def raft_prepare(sd, cfg, image1, image2):
    """Everything before the GRU loop (raft_stereo.py:91-114): returns
    fmap1, fmap2, net_list, inp_list (inp_list[i] = [cz, cr, cq])."""
    image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
    n = cfg['n_gru_layers']
    ds = cfg['n_downsample']
    cnet = multi_encoder(sd, 'cnet', image1, cfg['context_norm'], ds, n)
    both = basic_encoder(sd, 'fnet', torch.cat([image1, image2], dim=0), 'instance', ds)
    fmap1, fmap2 = both.split(image1.shape[0], dim=0)
    net = [torch.tanh(x[0]) for x in cnet]
    inp = [torch.relu(x[1]) for x in cnet]
    inp = [list(_conv(sd, 'context_zqr_convs.%d' % i, v).split(cfg['hidden_dims'][i], dim=1))
           for i, v in enumerate(inp)]
    return fmap1.float(), fmap2.float(), net, inp


def raft_iterations(sd, cfg, fmap1, fmap2, net, inp, iters, corr_impl='reg', flow_init=None,
                    trace=None):
    """The hot loop, raft_stereo.py:118-183 (test_mode=True): returns
    (coords1 - coords0, flow_up)."""
    n = cfg['n_gru_layers']
    L, r = cfg['corr_levels'], cfg['corr_radius']
    if corr_impl == 'reg':
        pyr = corr1d_pyramid(fmap1, fmap2, L)
        lookup = lambda c: corr1d_lookup(pyr, c, r)
    elif corr_impl == 'alt':
        lookup = lambda c: corr1d_lookup_alt(fmap1, fmap2, c, L, r)
    else:
        raise ValueError(corr_impl)
    b, _, h, w = net[0].shape
    coords0 = coords_grid(b, h, w).to(fmap1.device)
    coords1 = coords0.clone()
    if flow_init is not None:
        coords1 = coords1 + flow_init
    net = list(net)
    mask = None
    for _ in range(iters):
        corr = lookup(coords1)
        flow = coords1 - coords0
        if n == 3 and cfg.get('slow_fast_gru', False):
            net = update_block(sd, 'update_block', n, net, inp, it_coarse=True, it_mid=False, it_fine=False, update=False)
        if n >= 2 and cfg.get('slow_fast_gru', False):
            net = update_block(sd, 'update_block', n, net, inp, it_coarse=(n == 3), it_mid=True, it_fine=False, update=False)
        net, mask, delta = update_block(sd, 'update_block', n, net, inp, corr, flow,
                                        it_coarse=(n == 3), it_mid=(n >= 2))
        delta[:, 1] = 0.0
        coords1 = coords1 + delta
        if trace is not None:
            trace.append((coords1 - coords0)[:, :1].clone())
    flow_up = convex_upsample(coords1 - coords0, mask, 2 ** cfg['n_downsample'])[:, :1]
    return coords1 - coords0, flow_up


@torch.no_grad()
def raft_stereo_forward(sd, cfg, image1, image2, iters, corr_impl='reg'):
    fmap1, fmap2, net, inp = raft_prepare(sd, cfg, image1, image2)
    return raft_iterations(sd, cfg, fmap1, fmap2, net, inp, iters, corr_impl)


# --------------------------------------------------------------------------
# IGEV GRU loop from match features + geometry volume onward:
# meta_arch/igev_stereo/igev_stereo.py:192-210 (test_mode, without the
# timm-based feature network and spx upsampling, which cannot run offline).
# --------------------------------------------------------------------------
@torch.no_grad()
def igev_iterations(sd, cfg, match_left, match_right, geo_volume, init_disp, net, inp, iters):
    n = cfg['n_gru_layers']
    L, r = cfg['corr_levels'], cfg['corr_radius']
    gp, ip = geo_pyramids(match_left.float(), match_right.float(), geo_volume.float(), L)
    b, c, h, w = match_left.shape
    coords = torch.arange(w).float().to(match_left.device).reshape(1, 1, w, 1).repeat(b, h, 1, 1)
    disp = init_disp
    net = list(net)
    mask = None
    slow_fast = cfg.get('slow_fast_gru', False)
    for _ in range(iters):
        feat = geo_lookup(gp, ip, disp, coords, r)
        if n == 3 and slow_fast:       # igev_stereo.py:204-205: coarsest GRU alone
            net = update_block(sd, 'update_block', n, net, inp, it_coarse=True, it_mid=False, it_fine=False,
                               update=False, igev=True)
        if n >= 2 and slow_fast:       # igev_stereo.py:206-207: coarsest + middle GRU
            net = update_block(sd, 'update_block', n, net, inp, it_coarse=(n == 3), it_mid=True, it_fine=False,
                               update=False, igev=True)
        net, mask, delta = update_block(sd, 'update_block', n, net, inp, feat, disp,
                                        it_coarse=(n == 3), it_mid=(n >= 2), igev=True)
        disp = disp + delta
    return disp, mask


def epe(a, b):
    """tools/evaluate_stereo.py:149 (mean end-point error, 1-channel disparity)."""
    return torch.sum((a - b) ** 2, dim=1).sqrt().mean().item() if a.dim() == 4 else (a - b).abs().mean().item()


def fan_out_std(shape):
    return math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))


# --------------------------------------------------------------------------
# PCVNet correlation block: meta_arch/pcvnet/corr.py:18-61
# --------------------------------------------------------------------------
def pcv_pyramid(fmap1, fmap2, num_levels, downsample=2):
    """corr.py:19-31: all-pairs volume (scaled by 1/sqrt(C)) and num_levels-1 poolings by the
    compress factor (4 when downsample == 2, else 2)."""
    factor = 4 if downsample == 2 else 2
    vol = corr1d_volume(fmap1, fmap2)
    b, h, w1, _, w2 = vol.shape
    lvl = vol.reshape(b * h * w1, 1, 1, w2)
    pyr = [lvl]
    for _ in range(num_levels - 1):
        lvl = F.avg_pool2d(lvl, [1, factor], stride=[1, factor])
        pyr.append(lvl)
    return pyr, factor


def pcv_lookup(pyr, coords, sigma, sample_num, factor):
    """corr.py:33-51.  coords, sigma (B,G,H,W) -> (B, L*G*S, H, W)."""
    b, g, h, w = coords.shape
    n = b * h * w
    sg = sigma.permute(0, 2, 3, 1).contiguous().reshape(n, 1, g, 1)
    cx = coords.permute(0, 2, 3, 1).contiguous().reshape(n, 1, g, 1)
    half = sample_num // 2
    dx = torch.arange(-half, half + 1, dtype=torch.float32).view(1, 1, 1, sample_num)
    x = dx * sg + cx
    outs = []
    for i, lvl in enumerate(pyr):
        x0 = (x / factor ** i).reshape(n, 1, g * sample_num, 1)
        outs.append(_sample_rows(lvl.contiguous(), x0.contiguous()).view(b, h, w, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# --------------------------------------------------------------------------
# CGI normalised correlation volumes: meta_arch/cgi/submodule.py:143-180
# --------------------------------------------------------------------------
def _group_norm_corr(a, b, groups):
    bs, c, h, w = a.shape
    a = a.view(bs, groups, c // groups, h, w)
    b = b.view(bs, groups, c // groups, h, w)
    a = a / (torch.norm(a, 2, 2, True) + 1e-05)
    b = b / (torch.norm(b, 2, 2, True) + 1e-05)
    return (a * b).mean(dim=2)


def gwc_volume_norm(ref, tgt, maxdisp, groups):
    """build_gwc_volume_norm, submodule.py:154-164."""
    bs, c, h, w = ref.shape
    vol = ref.new_zeros(bs, groups, maxdisp, h, w)
    for d in range(maxdisp):
        if d == 0:
            vol[:, :, 0] = _group_norm_corr(ref, tgt, groups)
        else:
            vol[:, :, d, :, d:] = _group_norm_corr(ref[..., d:], tgt[..., :-d], groups)
    return vol.contiguous()


def norm_correlation_volume(ref, tgt, maxdisp):
    """build_norm_correlation_volume, submodule.py:167-180 (== igev_stereo/submodule.py:179)."""
    def nc(a, b):
        return torch.mean((a / (torch.norm(a, 2, 1, True) + 1e-05)) * (b / (torch.norm(b, 2, 1, True) + 1e-05)),
                          dim=1, keepdim=True)
    bs, c, h, w = ref.shape
    vol = ref.new_zeros(bs, 1, maxdisp, h, w)
    for d in range(maxdisp):
        if d == 0:
            vol[:, :, 0] = nc(ref, tgt)
        else:
            vol[:, :, d, :, d:] = nc(ref[..., d:], tgt[..., :-d])
    return vol.contiguous()


def context_upsample(disp_low, up_weights):
    """meta_arch/igev_stereo/submodule.py:242-254."""
    b, c, h, w = disp_low.shape
    nb = F.unfold(disp_low.reshape(b, c, h, w), 3, 1, 1).reshape(b, -1, h, w)
    nb = F.interpolate(nb, (h * 4, w * 4), mode='nearest').reshape(b, 9, h * 4, w * 4)
    return (nb * up_weights).sum(1)
