#!/usr/bin/env python3
"""Generate tests/golden/* from the reference's own fixtures (run in the authoring container, where
/root/reference exists; the outputs are committed because /root/reference does not exist on the GPU box).

Sources (all under /root/reference):
  aruco_detect/test/test_images/tag_01_d7_14cm.png, tag_245-246_d7_14cm.png
      golden ids + corners: aruco_detect/test/aruco_images_test.cpp:96-147 ; K/D :24-29
  fiducial_slam/test/test_images/403.jpg
      golden map pose: fiducial_slam/test/auto_init_403_test.cpp:129-137 ; base->camera TF auto_init_403.test:4
  fiducial_slam/test/aruco_images.bag  (frame seq 4957 + CameraInfo)
  fiducial_slam/test/aruco_transforms.bag (recorded FiducialTransformArray for the same seq)

Images are stored as the 8-bit gray the node's detector sees (cv_bridge BGR8 -> cvtColor BGR2GRAY,
OpenCV 4.x 15-bit fixed point, oracle.to_gray) plus a small colour crop for the bgr8/rgb8 input path.
"""
import io, json, os, struct, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/"
OUT = os.path.join(ROOT, "tests", "golden")


def parse_bag(path):
    """Minimal ROS bag v2.0 reader (uncompressed chunks). Returns {conn_id: topic/type}, [(conn, bytes)]."""
    data = open(path, "rb").read()
    assert data.startswith(b"#ROSBAG V2.0\n")
    conns, msgs = {}, []

    def records(buf, pos, end):
        while pos < end:
            hlen = struct.unpack_from("<I", buf, pos)[0]; pos += 4
            hdr = {}
            hend = pos + hlen
            while pos < hend:
                flen = struct.unpack_from("<I", buf, pos)[0]; pos += 4
                field = buf[pos:pos + flen]; pos += flen
                k, v = field.split(b"=", 1)
                hdr[k.decode()] = v
            dlen = struct.unpack_from("<I", buf, pos)[0]; pos += 4
            body = buf[pos:pos + dlen]; pos += dlen
            yield hdr, body

    def walk(buf, pos, end):
        for hdr, body in records(buf, pos, end):
            op = hdr["op"][0]
            if op == 5:  # chunk
                assert hdr["compression"] == b"none", hdr["compression"]
                walk(body, 0, len(body))
            elif op == 7:  # connection
                cid = struct.unpack("<I", hdr["conn"])[0]
                ch = {}
                for h2, _ in []:
                    pass
                # connection data is itself a header
                p = 0
                while p < len(body):
                    flen = struct.unpack_from("<I", body, p)[0]; p += 4
                    k, v = body[p:p + flen].split(b"=", 1); p += flen
                    ch[k.decode()] = v
                conns[cid] = (hdr["topic"].decode(), ch.get("type", b"").decode())
            elif op == 2:  # message
                cid = struct.unpack("<I", hdr["conn"])[0]
                msgs.append((cid, body))

    walk(data, 13, len(data))
    return conns, msgs


def rd_header(b, p):
    seq, sec, nsec, n = struct.unpack_from("<IIII", b, p); p += 16
    frame = b[p:p + n].decode(); p += n
    return dict(seq=seq, sec=sec, nsec=nsec, frame_id=frame), p


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {}

    def save_image(key, path, crop=None):
        rgb = np.asarray(Image.open(path).convert("RGB"))
        gray = oracle.to_gray(rgb, 2)
        arrs = {"gray": gray}
        if crop is not None:
            x0, y0, x1, y1 = crop
            arrs["rgb_crop"] = np.ascontiguousarray(rgb[y0:y1, x0:x1])
            arrs["crop_xyxy"] = np.array(crop, dtype=np.int32)
        np.savez_compressed(os.path.join(OUT, key + ".npz"), **arrs)
        print(key, gray.shape)

    # --- aruco_images_test.cpp -----------------------------------------------------------------
    save_image("tag_01", REF + "aruco_detect/test/test_images/tag_01_d7_14cm.png", crop=(480, 120, 864, 504))
    save_image("tag_245_246", REF + "aruco_detect/test/test_images/tag_245-246_d7_14cm.png", crop=(224, 96, 992, 480))
    meta["aruco_images_test"] = {
        "cite": "aruco_detect/test/aruco_images_test.cpp:96-147 (ASSERT_FLOAT_EQ = 4 ULP); K/D :24-29; dictionary 7, fiducial_len 0.145 (aruco_images.test:6)",
        "K": [1006.126285753055, 0.0, 655.8639244150409, 0.0, 1004.015433012594, 490.6140221242933, 0.0, 0.0, 1.0],
        "D": [0.1349735087283542, -0.2335869827451621, 0.0006697030315075139, 0.004846737465872353, 0.0],
        "fiducial_len": 0.145,
        "tag_01": {"1": [569.89917, 201.55890, 777.42560, 206.85025, 767.95856, 415.37830, 565.75311, 409.24496]},
        "tag_245_246": {
            "245": [307.68246, 157.38346, 545.10131, 167.04420, 540.11614, 403.27578, 305.64746, 395.01422],
            "246": [671.51892, 173.46070, 900.29650, 178.44973, 895.06933, 407.39855, 666.39910, 403.12911],
        },
    }
    # --- auto_init_403 ---------------------------------------------------------------------------
    save_image("img_403", REF + "fiducial_slam/test/test_images/403.jpg")
    meta["auto_init_403"] = {
        "cite": "fiducial_slam/test/auto_init_403_test.cpp:129-137 (ASSERT_NEAR 1e-3); auto_init_403.test:4,8; K/D auto_init_403_test.cpp (same CameraInfo literals as aruco_images_test)",
        "id": 403,
        "map_xyz": [0.7611, 0.2505, 0.4028],
        "map_rpy": [1.5751, -0.014, -1.546],
        "base_to_camera_xyz": [0.035, 0.145, 0.14],
        "base_to_camera_ypr": [-1.479119, -0.041544, -1.204205],
        "fiducial_len": 0.145,
    }
    # --- bag pair seq 4957 -------------------------------------------------------------------------
    conns, msgs = parse_bag(REF + "fiducial_slam/test/aruco_images.bag")
    K = D = None
    for cid, body in msgs:
        topic, typ = conns[cid]
        if typ == "sensor_msgs/CompressedImage":
            hdr, p = rd_header(body, 0)
            n = struct.unpack_from("<I", body, p)[0]; p += 4
            fmt = body[p:p + n].decode(); p += n
            n = struct.unpack_from("<I", body, p)[0]; p += 4
            jpg = body[p:p + n]
            rgb = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))
            gray = oracle.to_gray(rgb, 2)
            np.savez_compressed(os.path.join(OUT, "bag_4957.npz"), gray=gray)
            meta.setdefault("bag_4957", {}).update({"image_header": hdr, "format": fmt, "shape": list(gray.shape)})
            print("bag image", hdr, fmt, gray.shape)
        elif typ == "sensor_msgs/CameraInfo" and K is None:
            hdr, p = rd_header(body, 0)
            height, width = struct.unpack_from("<II", body, p); p += 8
            n = struct.unpack_from("<I", body, p)[0]; p += 4
            model = body[p:p + n].decode(); p += n
            n = struct.unpack_from("<I", body, p)[0]; p += 4
            D = list(struct.unpack_from("<%dd" % n, body, p)); p += 8 * n
            K = list(struct.unpack_from("<9d", body, p)); p += 72
            meta.setdefault("bag_4957", {}).update({"K": K, "D": D, "camera_frame": hdr["frame_id"], "distortion_model": model,
                                      "width": width, "height": height})
    conns, msgs = parse_bag(REF + "fiducial_slam/test/aruco_transforms.bag")
    for cid, body in msgs:
        topic, typ = conns[cid]
        if typ == "fiducial_msgs/FiducialTransformArray":
            hdr, p = rd_header(body, 0)
            image_seq = struct.unpack_from("<i", body, p)[0]; p += 4
            n = struct.unpack_from("<I", body, p)[0]; p += 4
            tfs = []
            for _ in range(n):
                fid = struct.unpack_from("<i", body, p)[0]; p += 4
                v = struct.unpack_from("<10d", body, p); p += 80
                tfs.append(dict(fiducial_id=fid, translation=list(v[0:3]), rotation_xyzw=list(v[3:7]),
                                image_error=v[7], object_error=v[8], fiducial_area=v[9]))
            meta["bag_4957"]["transforms"] = dict(
                cite="fiducial_slam/test/aruco_transforms.bag (recorded node output, 2017 / OpenCV 3.x era; JPEG decoder dependent)",
                header=hdr, image_seq=image_seq, transforms=tfs, raw_hex=body.hex())
            print("transforms", image_seq, [t["fiducial_id"] for t in tfs])
    json.dump(meta, open(os.path.join(OUT, "golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
