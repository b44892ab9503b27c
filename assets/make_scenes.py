"""Writes the .scene files of the benchmark configurations (assets/scenes/).

Geometry, materials, light and camera are those of the upstream Cornell
scenes (bin/cornell_box.scene, bin/cornell_sphere.scene,
bin/cornell_dragon.scene); the tracer block and image size are the benchmark
configurations: 200k global / 50k caustic photons, k = 100 / 50, depth 20,
roulette 0.8, radius (r^2) 0.08, 16 iterations.  The all-diffuse box keeps
csize = 0: with csize > 0 the upstream caustic loop never terminates there
(src/tracer/TkSppmTracer.cpp:465-470).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

MATERIALS = {
    "red": '<material name="red" type="diffuse"><R r="0.63" g="0.065" b="0.05"/></material>',
    "green": '<material name="green" type="diffuse"><R r="0.14" g="0.45" b="0.091"/></material>',
    "white": '<material name="white" type="diffuse"><R r="0.725" g="0.71" b="0.68"/></material>',
    "light": '<material name="light" type="diffuse"><R r="0.65" g="0.65" b="0.65"/></material>',
    "glass175": '<material name="glass" type="glass"><refractive_index v="1.75496"/>'
                '<R r="1.0" g="1.0" b="1.0"/><T r="1.0" g="1.0" b="1.0"/></material>',
    "glass22": '<material name="glass" type="glass"><refractive_index v="2.2"/>'
               '<R r="1.0" g="1.0" b="1.0"/><T r="1.0" g="1.0" b="1.0"/></material>',
    "mirror": '<material name="mirror" type="mirror"><R r="1.0" g="1.0" b="1.0"/></material>',
}


def mesh(name, file, pos=(-2.2, -2.2, 0.0), scale=0.008):
    return (f'<mesh name="{name}" filename="../models/{file}">'
            f'<position x="{pos[0]}" y="{pos[1]}" z="{pos[2]}"/>'
            f'<scale x="{scale}" y="{scale}" z="{scale}"/>'
            f'<orientation x="0.0" y="0.0" z="0.0"/></mesh>')


def scene(out, size, mats, meshes, objects, gsize, csize, recurse=16, threads=4):
    lines = [f'<scene filename="{out}">',
             f'<width i="{size}"/>', f'<height i="{size}"/>', '<windows i="0"/>',
             '<camera><fov v="40.0"/><near_clip v="0.1"/><far_clip v="100.0"/>'
             '<position x="-3.141592653" y="1.5707963265" z="6.5"/></camera>',
             f'<tracer type="sppm"><spp i="1"/><depth i="20"/><roulete v="0.8"/><thread i="{threads}"/>'
             f'<recurse i="{recurse}"/><gsize i="{gsize}"/><gsample i="100"/>'
             f'<csize i="{csize}"/><csample i="50"/><radius v="0.08"/></tracer>']
    lines += [MATERIALS[m] for m in mats]
    lines += meshes
    lines.append('<light name="light" mesh="light"><color r="47.8348" g="38.5664" b="31.0808"/></light>')
    lines += objects
    lines.append('</scene>')
    return "\n".join(lines) + "\n"


SHELL_MESHES = [mesh("floor", "cornellbox/floor.obj"), mesh("left", "cornellbox/left.obj"),
                mesh("right", "cornellbox/right.obj"), mesh("light", "cornellbox/light.obj")]
SHELL_OBJECTS = ['<object type="mesh" material="white" mesh="floor"/>',
                 '<object type="mesh" material="red" mesh="left"/>',
                 '<object type="mesh" material="green" mesh="right"/>',
                 '<object type="light" material="light" light="light" mesh="light"/>']


def main():
    out = os.path.join(HERE, "scenes")
    os.makedirs(out, exist_ok=True)
    box = scene("cornell_box.ppm", 512, ["red", "green", "white", "light"],
                SHELL_MESHES + [mesh("shortbox", "cornellbox/shortbox.obj"), mesh("tallbox", "cornellbox/tallbox.obj")],
                ['<object type="mesh" material="white" mesh="shortbox"/>',
                 '<object type="mesh" material="white" mesh="tallbox"/>'] + SHELL_OBJECTS,
                200000, 0)
    sphere_objs = ['<object type="sphere" material="glass"><position x="-0.68" y="-1.64" z="1.6"/><radius v="0.56"/></object>',
                   '<object type="sphere" material="mirror"><position x="0.6" y="-1.4" z="2.8"/><radius v="0.8"/></object>']
    sphere = scene("cornell_sphere.ppm", 512, ["red", "green", "white", "light", "glass175", "mirror"],
                   SHELL_MESHES, sphere_objs + SHELL_OBJECTS, 200000, 50000)
    dragon_meshes = SHELL_MESHES[:3] + [mesh("dragon", "dragon.obj", (0.0, 0.3, 2.5), 2.0)] + SHELL_MESHES[3:]
    dragon_objs = ['<object type="mesh" material="glass" mesh="dragon"/>'] + SHELL_OBJECTS
    dragon = scene("cornell_dragon.ppm", 1024, ["red", "green", "white", "light", "glass22"],
                   dragon_meshes, dragon_objs, 200000, 50000, threads=2)
    dragon8 = scene("cornell_dragon_2m.ppm", 1024, ["red", "green", "white", "light", "glass22"],
                    dragon_meshes, dragon_objs, 2000000, 500000, threads=2)
    for name, text in [("cornell_box.scene", box), ("cornell_sphere.scene", sphere),
                       ("cornell_dragon.scene", dragon), ("cornell_dragon_2m.scene", dragon8)]:
        with open(os.path.join(out, name), "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
