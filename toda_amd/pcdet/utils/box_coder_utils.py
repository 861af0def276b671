"""ResidualCoder (reference pcdet/utils/box_coder_utils.py:5-77): anchors <-> regression targets."""
import torch


class ResidualCoder:
    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        self.code_size = code_size + (1 if encode_angle_by_sincos else 0)
        self.encode_angle_by_sincos = encode_angle_by_sincos

    def encode_torch(self, boxes, anchors):
        """boxes, anchors [N, 7+C] -> targets [N, code_size]; sizes are clamped to >= 1e-5 in place
        (the reference mutates its inputs the same way)."""
        anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], min=1e-5)
        boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], min=1e-5)
        xa, ya, za, dxa, dya, dza, ra, *cas = torch.split(anchors, 1, dim=-1)
        xg, yg, zg, dxg, dyg, dzg, rg, *cgs = torch.split(boxes, 1, dim=-1)
        diag = torch.sqrt(dxa ** 2 + dya ** 2)
        parts = [(xg - xa) / diag, (yg - ya) / diag, (zg - za) / dza,
                 torch.log(dxg / dxa), torch.log(dyg / dya), torch.log(dzg / dza)]
        if self.encode_angle_by_sincos:
            parts += [torch.cos(rg) - torch.cos(ra), torch.sin(rg) - torch.sin(ra)]
        else:
            parts.append(rg - ra)
        parts += [g - a for g, a in zip(cgs, cas)]
        return torch.cat(parts, dim=-1)

    def decode_torch(self, box_encodings, anchors):
        xa, ya, za, dxa, dya, dza, ra, *cas = torch.split(anchors, 1, dim=-1)
        if self.encode_angle_by_sincos:
            xt, yt, zt, dxt, dyt, dzt, cost, sint, *cts = torch.split(box_encodings, 1, dim=-1)
        else:
            xt, yt, zt, dxt, dyt, dzt, rt, *cts = torch.split(box_encodings, 1, dim=-1)
        diag = torch.sqrt(dxa ** 2 + dya ** 2)
        out = [xt * diag + xa, yt * diag + ya, zt * dza + za,
               torch.exp(dxt) * dxa, torch.exp(dyt) * dya, torch.exp(dzt) * dza]
        if self.encode_angle_by_sincos:
            out.append(torch.atan2(sint + torch.sin(ra), cost + torch.cos(ra)))
        else:
            out.append(rt + ra)
        out += [t + a for t, a in zip(cts, cas)]
        return torch.cat(out, dim=-1)
